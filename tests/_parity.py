"""Shared parity checks: run the library (real HIP or emulator build) and the oracle on the same
inputs and compare bit for bit."""
import numpy as np

import _golden as G
import _oracle as O
from muscle_amd._lib import MpcGpu
from muscle_amd.synth import make_family


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def run_lib(seqs, iters=2, lib_path=None, hmm_name="hmm_amino", expf_variant=-1, mega=None, info=None):
    """mega = dict(alpha, weight, lp, mx, profs): stage A on structure profiles (mpcgpu_set_mega); info: a dict that receives
    stage_a_info"""
    s, t, m, i, thr = G.hmm_tables(hmm_name)
    g = MpcGpu(0, lib_path)
    g.set_hmm(s, t, m, i, thr, expf_variant)
    g.set_seqs(seqs)
    if mega is not None:
        g.set_mega(mega["alpha"], mega["weight"], mega["lp"], mega["mx"], mega["profs"])
    g.calc_posteriors()
    if info is not None:
        info["stage_a_info"] = g.stage_a_info()
    ea = g.get_ea().copy()
    g.build_store()
    stages = [g.get_sparse_range()]
    if len(seqs) >= 3:  # mpcflat.cpp:176
        for _ in range(iters):
            g.cons_iter()
            g.cons_commit()
            stages.append(g.get_sparse_range())
        if info is not None:
            info["relax_info"], info["relax_fallback"] = g.relax_info()
    g.close()
    return stages, ea


def random_mega(seqs, seed, nfeat=8):
    """Random structure-profile tables + profiles for the given sequences (library vs oracle runs: any finite
    negative log-probabilities will do; the reference-parsed ones are in the mega_* golden fixtures)."""
    rng = np.random.default_rng(seed)
    alpha = np.array([20] + [int(rng.choice([2, 5, 9, 16])) for _ in range(nfeat - 1)], np.uint32)[:nfeat]
    weight = rng.uniform(0.05, 0.5, nfeat).astype(np.float32)
    lp = np.concatenate([np.log(rng.dirichlet(np.ones(int(a)))).astype(np.float32) for a in alpha])
    mxs = []
    for a in alpha:
        a = int(a)
        j = rng.uniform(0.2, 1.0, (a, a)) + 5.0 * np.eye(a)
        j = (j + j.T) / 2
        mxs.append(np.log(j / j.sum()).astype(np.float32).ravel())
    amino = "ACDEFGHIKLMNPQRSTVWY"
    profs = []
    for sq in seqs:
        p = np.empty((len(sq), nfeat), np.uint8)
        for pos, c in enumerate(sq):
            base = amino.index(c) if c in amino else 0
            p[pos, 0] = base
            for f in range(1, nfeat):
                p[pos, f] = (base * 5 + f) % int(alpha[f]) if rng.random() > 0.3 else rng.integers(int(alpha[f]))
        profs.append(p.ravel())
    return {"alpha": alpha, "weight": weight, "lp": lp, "mx": np.concatenate(mxs), "profs": profs}


def run_oracle(seqs, iters=2, hmm_name="hmm_amino", threads=0, mega=None):
    s, t, m, i, thr = G.hmm_tables(hmm_name)
    h = O.make_hmm(s, t, m, i)
    if mega is not None:
        st = O.MegaStore(seqs, mega["profs"])
        ea = st.calc_posteriors_mega(h, O.make_mega(mega["alpha"], mega["weight"], mega["lp"], mega["mx"]), threads=threads)
    else:
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


def check_post_scores(lib_path=None, seed=11, trials=60):
    """mpcgpu_post_scores (both finishing kernels, one- and multi-pass EA) vs the dense CalcAlnScoreFlat / FromPost of the
    oracle on random candidate lists with what the pair-HMM rarely produces: empty rows and columns, rows whose first cell
    lies right of everything seen so far (a gap beyond the EA frontier), rows of more than 64 cells, 1-wide shapes."""
    rng = np.random.default_rng(seed)
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0, lib_path)
    variant = 1
    g.set_hmm(s, t, m, i, thr, variant)
    L = O.lib()
    shapes = [(1, 1), (1, 9), (7, 1), (3, 6), (12, 40), (40, 12), (30, 30), (5, 200), (90, 70)]
    # the advisor's minimal case: the dense DP gives 1.4
    fixed = [((3, 6), [(0, 3, 0.9), (1, 5, 0.3), (2, 2, 0.3), (2, 5, 0.5)])]
    cases = []
    for (LX, LY), cells in fixed:
        cases.append((LX, LY, [c[0] for c in cells], [c[1] for c in cells], np.log(np.array([c[2] for c in cells], np.float32))))
    for tr in range(trials):
        LX, LY = shapes[tr % len(shapes)]
        dens = [0.02, 0.08, 0.3, 0.9][tr % 4]
        mask = rng.random((LX, LY)) < dens
        if tr % 5 == 0:  # staircase with jumps: each row starts right of the previous frontier
            mask[:] = False
            c = 0
            for r in range(LX):
                c += int(rng.integers(1, 5))
                if c >= LY:
                    break
                mask[r, c] = True
                if rng.random() < 0.3 and c + 2 < LY:
                    mask[r, c + 2] = True
        rows, cols = np.nonzero(mask)
        perm = rng.permutation(len(rows))
        rows, cols = rows[perm], cols[perm]
        # scores from log(0.0098) (dropped by FromPost: P < 0.01) up to slightly above 0 (P = 1)
        sc = rng.uniform(np.log(0.0098), 0.05, len(rows)).astype(np.float32)
        sc = np.maximum(sc, np.float32(thr))
        cases.append((LX, LY, rows, cols, sc))
    for LX, LY, rows, cols, sc in cases:
        Pd = np.zeros((LX, LY), np.float32)
        for r, c, x in zip(rows, cols, sc):
            Pd[r, c] = np.float32(1.0) if x >= 0 else np.float32(L.orc_expf_emul(float(x), variant))
        want_score = O.aln_score(Pd)
        want_ea = np.float32(want_score) / np.float32(min(LX, LY))
        woff, wval = O.sparse_from_post(Pd)
        for kernel, batch in ((0, 64), (0, 3), (1, 64)):
            ea, off, val = g.post_scores(LX, LY, rows, cols, sc, kernel, batch)
            what = "%dx%d, %d cells, kernel %d batch %d" % (LX, LY, len(rows), kernel, batch)
            assert bits(ea) == bits(want_ea), "EA %s: %r vs %r" % (what, ea, want_ea)
            assert np.array_equal(off, woff), what
            assert np.array_equal(val, wval), what
    g.close()


BP_SETS = ["bp_n12_L70", "bp_n9_L40"]  # tests/golden/make_golden.py bp: matrices, paths, scores from the compiled reference


def check_buildpost_golden(name, lib_path=None):
    """Device BuildPost / AlignAlns / AlignMSAs against what the compiled reference produced for the same joins
    (tests/golden/<name>.npz): the C1 x C2 matrix of MPCFlat::BuildPost bit for bit (plain, weighted, both stored
    orientations), path and score of MPCFlat::AlignAlns; for explicit pair lists the matrix of CalcPosteriorFlat3, the path
    and the mean EA of PProg::AlignMSAsFlat's pieces."""
    import _buildpost as BP
    z = G.load(name)
    seqs = [str(x) for x in z["seqs"]]
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0, lib_path)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs(seqs)
    g.calc_posteriors()
    g.build_store()
    for _ in range(2):
        g.cons_iter()
        g.cons_commit()
    for j in range(int(z["njoins"])):
        k = "j%d_" % j
        grp1, grp2 = [int(x) for x in z[k + "idx1"]], [int(x) for x in z[k + "idx2"]]
        rows1, rows2 = [str(x) for x in z[k + "rows1"]], [str(x) for x in z[k + "rows2"]]
        C1, C2 = len(rows1[0]), len(rows2[0])
        m1, m2 = [BP.pos_to_col(r) for r in rows1], [BP.pos_to_col(r) for r in rows2]
        assert np.array_equal(bits(g.build_post(grp1, grp2, m1, m2, C1, C2)), bits(z[k + "post"])), (name, j, "matrix")
        w = z[k + "w"]  # the reference indexes m_Weights by the row number inside each alignment (buildpostflat.cpp:42,52)
        got = g.build_post(grp1, grp2, m1, m2, C1, C2, w[:len(grp1)], w[:len(grp2)])
        assert np.array_equal(bits(got), bits(z[k + "postw"])), (name, j, "weighted matrix")
        path, sc = g.align_alns(grp1, grp2, m1, m2, C1, C2)
        assert path == str(z[k + "path"]) and bits(sc) == bits(z[k + "score"]), (name, j, "path / score")
        assert np.array_equal(bits(g.last_post(C1, C2)), bits(z[k + "post"])), (name, j, "matrix after align_alns")
    g.close()
    g = MpcGpu(0, lib_path)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs_registry(seqs)
    for j in range(int(z["nmsas"])):
        k = "m%d_" % j
        grp1, grp2 = [int(x) for x in z[k + "idx1"]], [int(x) for x in z[k + "idx2"]]
        rows1, rows2 = [str(x) for x in z[k + "rows1"]], [str(x) for x in z[k + "rows2"]]
        C1, C2 = len(rows1[0]), len(rows2[0])
        r1, r2 = [int(x) for x in z[k + "row1"]], [int(x) for x in z[k + "row2"]]
        seq1, seq2 = [grp1[a] for a in r1], [grp2[b] for b in r2]
        m1, m2 = [BP.pos_to_col(rows1[a]) for a in r1], [BP.pos_to_col(rows2[b]) for b in r2]
        path, sc, ea = g.align_msas(seq1, seq2, m1, m2, C1, C2)
        assert path == str(z[k + "path"]), (name, j, "msas path")
        assert np.array_equal(bits(g.last_post(C1, C2)), bits(z[k + "post"])), (name, j, "msas matrix")
        tot = np.float32(0)
        for e in ea:  # getpostpairsalignedflat.cpp:90-96 with one thread: SumEA += EA in pair order, then / PairCount
            tot = np.float32(tot + np.float32(e))
        assert bits(np.float32(tot / np.float32(len(ea)))) == bits(z[k + "ea_avg"]), (name, j, "msas mean EA")
    g.close()


def check_align_pairs_golden(name="ap_ragged", lib_path=None):
    """mpcgpu_align_pairs (AlignPairFlat on the device: stage A on the pair list, dense thresholded posterior from the candidate
    lists, CalcAlnFlat + traceback, one launch for the batch) and mpcgpu_get_list_sparse against what the compiled reference's
    AlignPairFlat_SparsePost returns (tests/golden/ap_*.npz): path, EA bits, FromPost matrix — lengths 1..300, identical
    sequences, both index orders."""
    z = G.load(name)
    seqs = [str(x) for x in z["seqs"]]
    pairs = [(int(a), int(b)) for a, b in z["pairs"]]
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0, lib_path)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs_registry(seqs)
    res, sp = g.align_pairs([a for a, b in pairs], [b for a, b in pairs], sparse=True)
    for q, ((path, sc, ea), (off, val)) in enumerate(zip(res, sp)):
        assert path == str(z["p%d_path" % q]), (q, pairs[q])
        assert bits(ea) == bits(z["p%d_ea" % q]), (q, pairs[q])
        assert np.array_equal(off, z["p%d_off" % q]) and np.array_equal(val, z["p%d_val" % q]), (q, pairs[q])
    # one pair at a time gives the same answers (the drop-in's AlignPairFlat makes such calls)
    for q in (0, 3, 7):
        (path, sc, ea), = g.align_pairs([pairs[q][0]], [pairs[q][1]])
        assert path == str(z["p%d_path" % q]) and bits(ea) == bits(z["p%d_ea" % q])
    # lists of up to 64 pairs take the one-wait path of the library (nothing packed; the lists come from a second, general
    # stage on demand), longer ones the general stage: both must agree with the reference
    import os
    old = os.environ.get("MPCGPU_PAIRS_SMALL")
    try:
        for small in ("0", "1"):
            os.environ["MPCGPU_PAIRS_SMALL"] = small
            sel = list(range(min(len(pairs), 8)))
            res, sp = g.align_pairs([pairs[q][0] for q in sel], [pairs[q][1] for q in sel], sparse=True)
            for q, (path, sc, ea), (off, val) in zip(sel, res, sp):
                assert path == str(z["p%d_path" % q]) and bits(ea) == bits(z["p%d_ea" % q]), (small, q)
                assert np.array_equal(off, z["p%d_off" % q]) and np.array_equal(val, z["p%d_val" % q]), (small, q)
    finally:
        if old is None:
            os.environ.pop("MPCGPU_PAIRS_SMALL", None)
        else:
            os.environ["MPCGPU_PAIRS_SMALL"] = old
    g.close()


def check_fb_chains(lib_path=None):
    """fb_chain_kernel (kernels_fbc.h: consecutive pairs with the same row sequence swept back to back) against fb_kernel
    (MPCGPU_FB_CHAIN=0) and the oracle: lengths that give 1, 2 and 3 rows per lane, chains cut by the length rule (LY + 1 < T),
    by the chain limit (2, 3, 8 pairs) and by the end of a row's run of pairs; every stage snapshot and EA bit for bit."""
    import os
    from muscle_amd.synth import make_family
    fams = [make_family(7, 40, seed=21), make_family(6, 150, seed=22),
            make_family(3, 90, seed=23) + make_family(2, 30, seed=24) + make_family(3, 140, seed=25) + ["MKV", "ACDEFGHIKLMNPQRSTVWY" * 5]]
    old = {k: os.environ.get(k) for k in ("MPCGPU_FB_CHAIN", "MPCGPU_FB_CHAIN_MAX", "MPCGPU_FB_CHAIN_GRADE")}
    try:
        for seqs in fams:
            want = run_oracle(seqs)
            os.environ["MPCGPU_FB_CHAIN"] = "0"
            assert_same(run_lib(seqs, lib_path=lib_path), want, "fb_kernel")
            for cmax, grade in (("2", "0"), ("3", "0"), ("8", "0"), ("8", "1")):
                os.environ["MPCGPU_FB_CHAIN"] = "1"
                os.environ["MPCGPU_FB_CHAIN_MAX"] = cmax
                os.environ["MPCGPU_FB_CHAIN_GRADE"] = grade  # 1: shorter chains at the end of a launch (the default)
                info = {}
                got = run_lib(seqs, lib_path=lib_path, info=info)
                assert_same(got, want, "chains of up to %s" % cmax)
                if grade == "0":
                    assert info["stage_a_info"][1] >= 4 and info["stage_a_info"][2] >= 2, info  # chains did form
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def check_align_pairs_chunked(lib_path=None):
    """a list of 300 pairs (two stage-A chunks: 256 + 44) through mpcgpu_align_pairs + mpcgpu_get_list_sparse: every entry equals what the
    same ordered pair gives as a list of one — paths, score / EA bits, sparse matrices"""
    import _golden as G
    from muscle_amd._lib import MpcGpu
    from muscle_amd.synth import make_family
    seqs = make_family(6, 14, seed=21)
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0, lib_path) if lib_path else MpcGpu(0)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs(seqs)
    ordered = [(a, b) for a in range(6) for b in range(6) if a != b]
    lst = [ordered[q % len(ordered)] for q in range(300)]
    res, sp = g.align_pairs([a for a, _ in lst], [b for _, b in lst], sparse=True)
    assert len(sp) == 300
    one = {}
    for (a, b) in ordered:
        r1, s1 = g.align_pairs([a], [b], sparse=True)
        one[(a, b)] = (r1[0], s1[0])
    for q, (a, b) in enumerate(lst):
        (path, sc, ea), (off, val) = one[(a, b)]
        assert res[q][0] == path and bits(res[q][1]) == bits(sc) and bits(res[q][2]) == bits(ea), q
        assert np.array_equal(sp[q][0], off) and np.array_equal(sp[q][1], val), (q, a, b)
    g.close()


def check_full_alphabet(lib_path=None, n=5, length=64, seed=11):
    """Stage A + relax on sequences that use 127 distinct seven-bit byte values: the compacted emission tables are then
    (127 * 127 + 127) floats = 65 KB of LDS — beyond the 64 KB a kernel gets without asking (mpcgpu.cpp: ensure_dyn_smem, per device
    and function: round-5 advisor finding) — against the oracle, bit for bit."""
    rng = np.random.default_rng(seed)
    base = rng.integers(1, 128, size=length)
    seqs = []
    for _ in range(n):
        s = base.copy()
        flip = rng.random(length) < 0.25
        s[flip] = rng.integers(1, 128, size=int(flip.sum()))
        seqs.append(bytes(s.astype(np.uint8)))
    seqs[0] = bytes(range(1, 128))[:length] + seqs[0][length // 2:]  # every value 1..127 occurs somewhere
    seqs[1] = bytes(range(127, 0, -1))[:127 - length] + seqs[1][:length // 2] if length < 127 else seqs[1]
    assert len(set(b"".join(seqs))) >= 120
    got = run_lib(seqs, lib_path=lib_path)
    want = run_oracle(seqs)
    assert_same(got, want, "127-letter alphabet")


def check_align_alns_batch(lib_path=None, n=14, length=40, seed=29):
    """mpcgpu_align_alns_batch (the joins of one guide-tree level in two launches) against mpcgpu_align_alns join by join: same path,
    same score bits — small joins of 1..3 rows a side (the batched forms), wide ones (> 512 columns: 16 columns per lane), one that
    only the general form takes (forced through it: > 2048 pairs would need 46 x 46 rows), and a batch of one."""
    import _buildpost as BP
    rng = np.random.default_rng(seed)
    seqs = make_family(n - 2, length, seed=seed) + make_family(2, 300, seed=seed + 1)
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0, lib_path)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs(seqs)
    g.calc_posteriors()
    g.build_store()
    for _ in range(2):
        g.cons_iter()
        g.cons_commit()
    groups = [([0], [1]), ([2, 3], [4]), ([5, 6, 7], [8, 9]), ([10], [11, 0]), ([n - 2], [n - 1]), ([n - 1, 3], [n - 2]), ([1, 2, 4], [5, 9, 10])]
    joins = []
    for grp1, grp2 in groups:
        rows1, C1 = BP.random_msa(seqs, grp1, rng)
        rows2, C2 = BP.random_msa(seqs, grp2, rng)
        if n - 1 in grp1 + grp2:  # the long sequences, spread over > 512 columns on the MSA2 side
            rows2, C2 = BP.random_msa(seqs, grp2, rng, extra=300)
        joins.append((grp1, grp2, [BP.pos_to_col(r) for r in rows1], [BP.pos_to_col(r) for r in rows2], C1, C2))
    want = [g.align_alns(*j) for j in joins]
    got = g.align_alns_batch(joins)
    for q, ((p0, s0), (p1, s1)) in enumerate(zip(want, got)):
        assert p0 == p1 and bits(s0) == bits(s1), ("join", q, groups[q])
    (p1, s1), = g.align_alns_batch(joins[2:3])  # a batch of one takes the single-join path
    assert p1 == want[2][0] and bits(s1) == bits(want[2][1])
    g.close()
