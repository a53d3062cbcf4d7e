"""Pins the CPU oracle (oracle/mpc_oracle.c) bit-for-bit against the compiled reference
(oracle/_ref/libmuscle_ref.so = /root/reference/src built by oracle/build_ref.sh).
Skipped where the compiled reference is absent (then tests/test_oracle_golden.py is the pin)."""
import random

import numpy as np
import pytest

import _oracle as O
import _ref as R
from muscle_amd.synth import make_family, AMINO

pytestmark = pytest.mark.skipif(not R.available(), reason="compiled reference not present")

# the reference's disabled unit test's short pairs (testfb.cpp:369-375)
SHORT = [("MQTIF", "MSIF"), ("GATTACA", "MQTIF"), ("ABC", "DEF"),
         ("LQNGSEQVENCE", "QTHERSEQVENCEINSERT")]
EDGE = [("A", "A"), ("A", "ACDEFGHIKL"), ("ACDEFGHIKLMNPQRSTVWY" * 3, "W"), ("XXBZ", "AXCB"),
        ("acdef", "ACDEF"), ("MKV", "MKV")]


def beq(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.fixture(scope="module")
def hmm():
    R.init_hmm(False, 0)
    return O.make_hmm(*R.get_hmm())


def rand_pairs():
    rng = random.Random(7)
    out = []
    fam = make_family(6, 120, seed=3)
    out += [(fam[0], fam[1]), (fam[2], fam[5])]
    for _ in range(6):
        out.append(("".join(rng.choice(AMINO) for _ in range(rng.randint(1, 90))),
                    "".join(rng.choice(AMINO) for _ in range(rng.randint(1, 90)))))
    return out


@pytest.mark.parametrize("x,y", SHORT + EDGE + rand_pairs())
def test_pair_pipeline_bit_exact(hmm, x, y):
    LX, LY = len(x), len(y)
    Fo, Fr = O.fwd(hmm, x, y), R.fwd(x, y)
    assert beq(Fo, Fr)
    Bo, Br = O.bwd(hmm, x, y), R.bwd(x, y)
    assert beq(Bo, Br)
    to, tr = O.total(Fo, Bo, LX, LY), R.total(Fr, Br, LX, LY)
    assert np.float32(to).view(np.uint32) == np.float32(tr).view(np.uint32)
    Po, Pr = O.post(Fo, Bo, LX, LY), R.post(Fr, Br, LX, LY)
    assert beq(Po, Pr)
    (oo, vo), (orr, vr) = O.sparse_from_post(Po), R.sparse_from_post(Pr)
    assert np.array_equal(oo, orr) and np.array_equal(vo, vr)
    assert np.float32(O.aln_score(Po)).view(np.uint32) == np.float32(R.aln_score(Pr)).view(np.uint32)
    so, po = O.calc_aln(Po)
    sr, pr = R.calc_aln(Pr)
    assert po == pr and np.float32(so).view(np.uint32) == np.float32(sr).view(np.uint32)


def test_min_sparse_score(hmm):
    assert np.float32(O.lib().orc_min_sparse_score()).view(np.uint32) == \
        np.float32(R.lib().ref_min_sparse_score()).view(np.uint32)


def test_calc_aln_dense_random():
    # progressive-alignment style input: arbitrary dense matrices incl. ties (best3.h tie order)
    rng = np.random.default_rng(5)
    for (a, b) in [(1, 1), (1, 7), (9, 1), (13, 17), (40, 33)]:
        P = (rng.integers(0, 4, size=(a, b)) / 4.0).astype(np.float32)
        so, po = O.calc_aln(P)
        sr, pr = R.calc_aln(P)
        assert po == pr and so == sr
        assert O.aln_score(P) == R.aln_score(P)


def _full_two_iterations(args):
    seqs, = args
    import hashlib
    stages, _ = R.mpc_run(seqs, iters=2, threads=2)
    return [hashlib.sha256(o.tobytes() + v.tobytes()).digest() for o, v in stages[2]]


def test_stage2_clique_shortcut_equals_two_full_iterations():
    """tests/golden/make_golden.py big-stage2 (the stage-2 pin of rdrp-1000, whose full relax takes days on the CPU) relaxes only what
    two ConsPair of a clique's pairs read: ConsPair of iteration 1 for every pair that touches the clique, the swap of consflat.cpp:22,
    ConsPair of iteration 2 for the pairs among the clique. On 14 sequences the clique's stage-2 matrices must be those of two FULL
    ConsIter of the same reference (each run in its own process: one MPCFlat per process)."""
    import multiprocessing as mp
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as MG
    n, m = 14, 4
    seqs = make_family(n, 60, seed=9)
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        full = pool.map(_full_two_iterations, [(seqs,)])[0]
    with ctx.Pool(1) as pool:
        d = pool.map(MG._mpc_stage2_worker, [("synth", n, m, 2, True)])[0]
    assert len(d["stage2_k"]) == m * (m - 1) // 2
    for q, k in enumerate(d["stage2_k"]):
        assert d["stage2_sha"][q].tobytes() == full[int(k)], (q, int(k))
