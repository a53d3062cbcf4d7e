"""Build-time checks of the compiled gfx950 code of the hot kernels (hipcc cross-compiles without a GPU): what the register
allocator did is a performance property that parity tests cannot see.

relax_var_kernel walks Z with every accumulator and row offset in VGPRs; a value the compiler keeps in a spill slot instead is
reloaded between the merges of a step, and each reload waits for vmcnt(0). The 14-cells-per-lane instantiation of the default
geometry had five such reloads (1195 ms per two iterations at 1000 x L~400 against 1171 with 13 cells per lane and none:
profiles/r05g). The forward/backward kernels must not spill at all."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "muscle_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "mpcgpu.gfx950.s"
    flags = "-O3 -fno-slp-vectorize -std=c++17 -fPIC -ffp-contract=off".split()  # the Makefile's CXXFLAGS
    subprocess.check_call([HIPCC, "-x", "hip", "--offload-arch=gfx950"] + flags + ["-S", "--cuda-device-only", "mpcgpu.cpp", "-o", str(out)],
                          cwd=CSRC, stderr=subprocess.DEVNULL)
    return open(out).read()


def _body(isa, mangled_prefix):
    i = isa.index(mangled_prefix)
    i = isa.index(":\n", i)
    return isa[i:isa.index(".Lfunc_end", i)].split("\n")


def _scratch(lines):
    return [k for k, l in enumerate(lines) if re.search(r"\bscratch_(load|store)|\bbuffer_(load|store)", l)]


def test_default_relax_walk_has_no_spill_reloads(isa):
    # the instantiation relax_var_launch picks by default: var_slots_2048() == 13
    src = open(os.path.join(CSRC, "mpcgpu.cpp")).read()
    assert re.search(r'env_int\("MPCGPU_RELAX_SLOTS_2048", 13\)', src), "default cells per lane of the default geometry changed: update this test"
    body = _body(isa, "_Z16relax_var_kernelILi1024ELi13ELi2ELi0E14MpcRvBlocksAsmEv14RelaxVarParams")
    merges = [k for k, l in enumerate(body) if re.match(r"\.Lrv_step_\d+:", l.strip())]
    assert len(merges) == 13, len(merges)  # one hand-scheduled merge loop per cell slot
    inside = [k for k in _scratch(body) if merges[0] <= k <= merges[-1]]
    assert not inside, "spill code between the merges of a step: " + "; ".join(body[k].strip() for k in inside[:5])


def test_forward_backward_kernels_do_not_spill(isa):
    for name in ("_Z15fb_chain_kernelILi7EEv13FbChainParams", "_Z9fb_kernelILi7ELb0ELb0EEv8FbParams"):
        body = _body(isa, name)
        assert not _scratch(body), name
