"""Run by tests/test_emu_parity.py::test_emu_other_thread_orders in a fresh process with EMU_SCHED=reverse / random: the
SIMT emulator then hands the OS thread to the GPU threads of a block in another order than 0..n-1 (tests/emu/hip_emu.cpp).
Between two barriers a GPU thread runs uninterrupted, so a missing barrier or any other race inside a barrier interval
makes the result depend on that order; these scenarios cover every kernel that synchronises through LDS or global
memory within a workgroup (fb with row blocks, post, pad build, both relax layouts incl. the two-buffer schedule,
calc_aln, the BuildPost path of align_alns)."""
import os
import sys

import numpy as np

import _parity as P
from muscle_amd.synth import make_family

EMU = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libmpcgpu_emu.so")


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    seqs = make_family(9, 18, seed=5) + [make_family(1, 70, seed=9)[0], "MKV"]
    want = P.run_oracle(seqs)
    for env in ({}, {"MPCGPU_RELAX_WG": "1024"}, {"MPCGPU_RELAX_WG": "1024", "MPCGPU_RELAX_NBUF": "1"}, {"MPCGPU_RELAX": "gather"},
                {"MPCGPU_POST": "sort"}, {"MPCGPU_RELAX_WG": "512"}):
        P.assert_same(with_env(env, lambda: P.run_lib(seqs, lib_path=EMU)), want, "order %s %s" % (os.environ.get("EMU_SCHED"), env))
    seqs = [make_family(1, 131, seed=21)[0], make_family(1, 66, seed=22)[0], "MKV"]
    got = with_env({"MPCGPU_FB_LONG_H": "1", "MPCGPU_FB_LONG_MIN": "2"}, lambda: P.run_lib(seqs, lib_path=EMU))
    P.assert_same(got, P.run_oracle(seqs), "row blocks")
    # calc_aln (prefix-max scan through LDS) on a matrix with many ties
    import _golden as G
    import _oracle as O
    from muscle_amd._lib import MpcGpu
    g = MpcGpu(0, EMU)
    rng = np.random.default_rng(3)
    M = ((rng.random((40, 300)) < 0.05) * rng.integers(1, 4, (40, 300)) * 0.25).astype(np.float32)
    path, sc = g.calc_aln(M)
    osc, opath = O.calc_aln(M)
    assert path == opath and np.float32(sc).view(np.uint32) == np.float32(osc).view(np.uint32)
    g.close()
    print("OK thread order", os.environ.get("EMU_SCHED", "forward"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
