"""Shared parity checks: run the library (real HIP or emulator build) and the oracle on the same
inputs and compare bit for bit."""
import numpy as np

import _golden as G
import _oracle as O
from muscle_amd._lib import MpcGpu


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def run_lib(seqs, iters=2, lib_path=None, hmm_name="hmm_amino", expf_variant=-1):
    s, t, m, i, thr = G.hmm_tables(hmm_name)
    g = MpcGpu(0, lib_path)
    g.set_hmm(s, t, m, i, thr, expf_variant)
    g.set_seqs(seqs)
    g.calc_posteriors()
    ea = g.get_ea().copy()
    g.build_store()
    stages = [g.get_sparse_range()]
    if len(seqs) >= 3:  # mpcflat.cpp:176
        for _ in range(iters):
            g.cons_iter()
            g.cons_commit()
            stages.append(g.get_sparse_range())
    g.close()
    return stages, ea


def run_oracle(seqs, iters=2, hmm_name="hmm_amino", threads=0):
    s, t, m, i, thr = G.hmm_tables(hmm_name)
    h = O.make_hmm(s, t, m, i)
    st = O.Store(seqs)
    ea = st.calc_posteriors(h, threads=threads)
    stages = [[st.get(k) for k in range(st.npairs)]]
    cur = st
    if len(seqs) >= 3:
        for _ in range(iters):
            cur = cur.cons_iter(threads=threads)
            stages.append([cur.get(k) for k in range(st.npairs)])
    return stages, ea


def assert_same(a, b, what=""):
    (sa, ea), (sb, eb) = a, b
    assert np.array_equal(bits(ea), bits(eb)), "EA differs %s: %s vs %s" % (what, ea[:5], eb[:5])
    assert len(sa) == len(sb)
    for s, (x, y) in enumerate(zip(sa, sb)):
        assert len(x) == len(y)
        for k, ((o1, v1), (o2, v2)) in enumerate(zip(x, y)):
            assert np.array_equal(o1, o2), "%s stage %d pair %d offsets" % (what, s, k)
            assert np.array_equal(v1, v2), "%s stage %d pair %d values" % (what, s, k)
