"""CPU-side checks of the C ABI: the built library loads and exports every symbol include/mpcgpu.h
declares (no compute calls without a GPU), and refuses to run without a device."""
import os
import re

import pytest

from muscle_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mpcgpu.h")).read()
    return sorted(set(re.findall(r"\b(mpcgpu_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_symbol():
    if not os.path.exists(_lib.DEFAULT_LIB):
        import __graft_entry__ as ge
        ge.build()
    L = _lib.load()
    for s in declared_symbols():
        assert hasattr(L, s), s
    assert b"gfx950" in L.mpcgpu_version()


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.MpcGpuError) as ei:
        _lib.MpcGpu(0)
    assert "no HIP device" in str(ei.value) or "no CPU path" in str(ei.value)
