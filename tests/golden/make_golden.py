#!/usr/bin/env python3
"""Generates the golden fixtures in tests/golden/ FROM THE COMPILED REFERENCE
(oracle/_ref/libmuscle_ref.so, built from /root/reference/src by oracle/build_ref.sh).

Run here (the container that has /root/reference):   python tests/golden/make_golden.py
The GPU box has no /root/reference; it only reads the committed .npz files.

Fixtures
  hmm_amino.npz / hmm_nucleo.npz : PairHMM tables after HMMParams::FromDefaults + ToPairHMM
                                   (hmmparams.cpp:273,298) and MIN_SPARSE_SCORE (mysparsemx.h:4)
  pairs_small.npz                : per-pair outputs of CalcFwdFlat/CalcBwdFlat/CalcTotalProbFlat/
                                   CalcPostFlat/FromPost/CalcAlnScoreFlat/CalcAlnFlat for the short
                                   pairs of the reference's disabled unit test (testfb.cpp:369-375)
                                   and edge cases; F/B M-planes in full, sha256 of the full 5-state arrays
  mega_<name>.npz                : the same stage for a .mega input (structure profiles, calcpost.cpp:14-22):
                                   the reference's parsed Mega tables + profiles, the stage snapshots in
                                   full, and probes of Mega::GetInsScore/GetMatchScore/CalcFwdFlat_mega/
                                   CalcBwdFlat_mega on the first pair
  mpc_<name>.npz                 : whole stage through the reference's MPCFlat::CalcPosteriors +
                                   ConsIter x2 (one subprocess per data set): EA per pair and, per
                                   stage, the MySparseMx arrays in full (small sets) or their sha256
                                   digests (larger sets)
"""
import hashlib
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHORT = [("MQTIF", "MSIF"), ("GATTACA", "MQTIF"), ("ABC", "DEF"),
         ("LQNGSEQVENCE", "QTHERSEQVENCEINSERT")]
EDGE = [("A", "A"), ("A", "ACDEFGHIKL"), ("ACDEFGHIKLMNPQRSTVWY" * 3, "W"), ("XXBZ", "AXCB"),
        ("acdef", "ACDEF"), ("MKV", "MKV")]


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def stage_digest(stage):
    h = hashlib.sha256()
    for off, val in stage:
        h.update(off.tobytes())
        h.update(val.tobytes())
    return h.hexdigest()


def gen_hmm():
    import _ref as R
    for name, nuc in (("hmm_amino", False), ("hmm_nucleo", True)):
        R.init_hmm(nuc, 0)
        start, trans, match, ins = R.get_hmm()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), start=start, trans=trans,
                            match=match.reshape(256, 256), ins=ins,
                            min_sparse_score=np.float32(R.lib().ref_min_sparse_score()))
    R.init_hmm(False, 0)


def gen_pairs_small():
    import _ref as R
    from muscle_amd.synth import make_family
    R.init_hmm(False, 0)
    fam = make_family(4, 70, seed=11)
    pairs = SHORT + EDGE + [(fam[0], fam[1]), (fam[2], fam[3])]
    d = {"n": np.int32(len(pairs))}
    for k, (x, y) in enumerate(pairs):
        LX, LY = len(x), len(y)
        F, B = R.fwd(x, y), R.bwd(x, y)
        P = R.post(F, B, LX, LY)
        off, val = R.sparse_from_post(P)
        sc, path = R.calc_aln(P)
        d["x%d" % k] = np.frombuffer(x.encode(), np.uint8)
        d["y%d" % k] = np.frombuffer(y.encode(), np.uint8)
        d["FM%d" % k] = F.reshape(LX + 1, LY + 1, 5)[:, :, 0].copy()
        d["BM%d" % k] = B.reshape(LX + 1, LY + 1, 5)[:, :, 0].copy()
        d["Fsha%d" % k] = np.array(sha(F))
        d["Bsha%d" % k] = np.array(sha(B))
        d["total%d" % k] = np.float32(R.total(F, B, LX, LY))
        d["post%d" % k] = P
        d["off%d" % k] = off
        d["val%d" % k] = val
        d["alnscore%d" % k] = np.float32(R.aln_score(P))
        d["calcaln_score%d" % k] = np.float32(sc)
        d["path%d" % k] = np.array(path)
    np.savez_compressed(os.path.join(HERE, "pairs_small.npz"), **d)


def read_fasta(path):
    import gzip
    seqs, cur = [], []
    for line in (gzip.open(path, "rt") if path.endswith(".gz") else open(path)):
        line = line.strip()
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur))
            cur = []
        elif line:
            cur.append(line.upper())  # sequence.cpp:87-88 upper-cases at load
    if cur:
        seqs.append("".join(cur))
    return seqs


def _mpc_worker(args):
    name, seqs, full = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _ref as R
    stages, ea = R.mpc_run(seqs, iters=2, threads=0)
    d = {"seqs": np.array(seqs), "ea": ea, "nstages": np.int32(len(stages))}
    for s, st in enumerate(stages):
        d["digest%d" % s] = np.array(stage_digest(st))
        d["nnz%d" % s] = np.array([len(v) // 2 for _, v in st], np.uint32)
        if full:
            d["off%d" % s] = np.concatenate([o for o, _ in st])
            d["val%d" % s] = np.concatenate([v for _, v in st]) if st else np.zeros(0, np.uint32)
    np.savez_compressed(os.path.join(HERE, "mpc_%s.npz" % name), **d)
    return name, len(seqs), [int(sum(len(v) // 2 for _, v in st)) for st in stages]


def odd_alphabet_family():
    """5 sequences of L~70 in which a third of the residues are replaced by arbitrary printable bytes (lower case, digits,
    punctuation): 82 distinct byte values — more than the 64 a compacted LDS table of the first rounds took."""
    import random
    from muscle_amd.synth import make_family
    rnd = random.Random(11)
    odd = [chr(b) for b in range(33, 127) if chr(b) not in ">-." and not chr(b).isupper() or chr(b) in "BJOUXZ"]
    seqs = []
    for s in make_family(5, 70, seed=4):
        t = list(s)
        for i in range(len(t)):
            if rnd.random() < 0.35:
                t[i] = rnd.choice(odd)
        seqs.append("".join(t))
    assert len(set("".join(seqs))) == 82
    return seqs


def gen_mpc(only=None):
    from muscle_amd.synth import make_family
    jobs = [
        ("n2_L40", make_family(2, 40, seed=5), True),       # N<3: consistency skipped (mpcflat.cpp:176)
        ("n3_L30", make_family(3, 30, seed=6), True),
        ("n8_L60", make_family(8, 60, seed=2), True),
        ("ragged", ["M", "MKVLA", make_family(1, 90, seed=9)[0], "ACDEFGHIKLMNPQRSTVWY" * 2,
                    make_family(1, 33, seed=10)[0], "WWWWWWWW"], True),
        ("alpha82", odd_alphabet_family(), True),           # 82 distinct byte values: the reference indexes its tables by raw byte (pairhmm.h:28-29)
        ("n32_L150", make_family(32, 150, seed=1), False),  # BASELINE config 0 shape
        ("n48_L260", make_family(48, 260, seed=1), False),
        ("n3_L1100", make_family(3, 1100, seed=7), False),  # X longer than 1024: the row-block fb kernel's shape
    ]
    bb = "/root/reference/test_data/fa/BB11001"
    if os.path.exists(bb):
        jobs.append(("bb11001", read_fasta(bb), True))
    bb5 = "/root/reference/test_data/fa/BB11005"
    if os.path.exists(bb5):
        jobs.append(("bb11005", read_fasta(bb5), False))
    # real data with wide posterior rows (r ~ 6 stored cells per row against ~2 on the synthetic family): the first 128 records of
    # the reference's test_data/rdrp/rdrp.fa (its first 1000 records are committed as rdrp_first1000.fa.gz for bench.py --fasta)
    rd = os.path.join(HERE, "rdrp_first1000.fa.gz")
    if os.path.exists(rd):
        jobs.append(("rdrp128", read_fasta(rd)[:128], False))
    ctx = mp.get_context("spawn")
    if only:
        jobs = [j for j in jobs if j[0] in only]
    for job in jobs:  # one process per data set: SetGlobalInputMS is once-per-process
        with ctx.Pool(1) as pool:
            print(pool.map(_mpc_worker, [job])[0])


BLOCK = 1000  # pairs per block digest of the digest-only fixtures


def _mpc_big_worker(args):
    """Digest-only fixture of a BASELINE-size set (configs 2 and 3): nothing is kept per pair but its nnz is
    folded into per-block sha256 digests (BLOCK pairs each, InitPairs order), so a mismatch on the GPU box
    localises to 1000 pairs; the whole-stage digest is the same stage_digest() as the small sets'."""
    name, n, length, seed, threads = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import time
    import _ref as R
    from muscle_amd.synth import make_family, read_fasta
    if length == 0:  # real data: the first n records of the reference's test_data/rdrp/rdrp.fa (tests/golden/rdrp_first1000.fa.gz)
        seqs = read_fasta(os.path.join(HERE, "rdrp_first1000.fa.gz"))[:n]
        assert len(seqs) == n
    else:
        seqs = make_family(n, length, seed=seed)
    R.init_hmm(False, 0)
    L = R.lib()
    arr = (C.c_char_p * len(seqs))(*[s.encode() for s in seqs])
    if L.ref_mpc_begin(len(seqs), arr, threads) != 0:
        raise RuntimeError("ref_mpc_begin may only be called once per process")
    lens = [len(s) for s in seqs]
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    u32p, u8p = R.u32p, R.u8p

    def snap():
        whole, blocks, blk, tot = hashlib.sha256(), [], hashlib.sha256(), 0
        for k, (i, j) in enumerate(pairs):
            nnz = L.ref_mpc_nnz(k)
            off = np.empty(lens[i] + 1, np.uint32)
            val = np.empty(max(nnz, 1) * 2, np.uint32)
            L.ref_mpc_sparse(k, off.ctypes.data_as(u32p), val.ctypes.data_as(u8p))
            b = off.tobytes() + val[:2 * nnz].tobytes()
            whole.update(b)
            blk.update(b)
            tot += nnz
            if (k + 1) % BLOCK == 0 or k + 1 == len(pairs):
                blocks.append(blk.digest())
                blk = hashlib.sha256()
        return whole.hexdigest(), np.frombuffer(b"".join(blocks), np.uint8).reshape(-1, 32), tot

    t0 = time.time()
    L.ref_mpc_calc_posteriors()
    tA = time.time() - t0
    ea = np.array([L.ref_mpc_ea(i, j) for (i, j) in pairs], np.float32)
    eab = [hashlib.sha256(ea[b:b + BLOCK].tobytes()).digest() for b in range(0, len(ea), BLOCK)]
    d = {"n": np.int32(n), "length": np.int32(length), "seed": np.int32(seed), "block": np.int32(BLOCK),
         "seqs_sha": np.array(hashlib.sha256("\n".join(seqs).encode()).hexdigest()),
         "ea_sha": np.array(hashlib.sha256(ea.tobytes()).hexdigest()),
         "ea_blocks": np.frombuffer(b"".join(eab), np.uint8).reshape(-1, 32)}
    stages = [snap()]
    tB = []
    for it in range(2):
        t0 = time.time()
        L.ref_mpc_cons_iter(it)
        tB.append(time.time() - t0)
        stages.append(snap())
    d["nstages"] = np.int32(len(stages))
    for s, (dig, blocks, tot) in enumerate(stages):
        d["digest%d" % s] = np.array(dig)
        d["blocks%d" % s] = blocks
        d["nnz_total%d" % s] = np.int64(tot)
    d["ref_seconds"] = np.array([tA] + tB)
    d["ref_threads"] = np.int32(threads)
    np.savez_compressed(os.path.join(HERE, "mpcbig_%s.npz" % name), **d)
    return name, n, [st[2] for st in stages], [tA] + tB


# BASELINE configs 2 and 3 (bench.py's families), and real data (length 0 = first n rdrp records) large enough for the
# relax tile mix of the 1000-record run (records of tens of KB: 4x1 / 2x1 tiles in the 160 KB geometry)
BIG_SETS = {"n256_L300": (256, 300, 1), "n1000_L400": (1000, 400, 1), "rdrp256": (256, 0, 0), "rdrp384": (384, 0, 0)}


def gen_mpc_big(names, threads):
    ctx = mp.get_context("spawn")
    for name in names:
        n, length, seed = BIG_SETS[name]
        with ctx.Pool(1) as pool:
            print(pool.map(_mpc_big_worker, [(name, n, length, seed, threads)])[0], flush=True)


def _mpc_sampled_worker(args):
    """SAMPLED pin of a store whose full ConsIter takes days on the CPU (rdrp, N = 1000: RelaxFlat_XZ_YZ's linear search at ~7
    stored cells per row): the compiled reference runs stage A for ALL pairs (EA + stage-0 digests per block of 1000 pairs, as the
    full fixtures hold them) and MPCFlat::ConsPair (conspairflat.cpp:10-110) of iteration 1 for `nsample` seeded pairs; per sampled
    pair the sha256 of (offsets || values) of its UPDATED matrix. The store is not swapped (no iteration 2)."""
    name, n, nsample, threads = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import time
    import _ref as R
    from muscle_amd.synth import read_fasta
    seqs = read_fasta(os.path.join(HERE, "rdrp_first1000.fa.gz"))[:n]
    assert len(seqs) == n
    R.init_hmm(False, 0)
    L = R.lib()
    L.ref_mpc_updated_nnz.restype = C.c_uint
    arr = (C.c_char_p * len(seqs))(*[s.encode() for s in seqs])
    if L.ref_mpc_begin(len(seqs), arr, threads) != 0:
        raise RuntimeError("ref_mpc_begin may only be called once per process")
    lens = [len(s) for s in seqs]
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    u32p, u8p = R.u32p, R.u8p
    t0 = time.time()
    L.ref_mpc_calc_posteriors()
    tA = time.time() - t0
    print("stage A: %.1f s" % tA, flush=True)
    ea = np.array([L.ref_mpc_ea(i, j) for (i, j) in pairs], np.float32)
    eab = [hashlib.sha256(ea[b:b + BLOCK].tobytes()).digest() for b in range(0, len(ea), BLOCK)]
    whole, blocks, blk, tot = hashlib.sha256(), [], hashlib.sha256(), 0
    for k, (i, j) in enumerate(pairs):
        nnz = L.ref_mpc_nnz(k)
        off = np.empty(lens[i] + 1, np.uint32)
        val = np.empty(max(nnz, 1) * 2, np.uint32)
        L.ref_mpc_sparse(k, off.ctypes.data_as(u32p), val.ctypes.data_as(u8p))
        b = off.tobytes() + val[:2 * nnz].tobytes()
        whole.update(b); blk.update(b); tot += nnz
        if (k + 1) % BLOCK == 0 or k + 1 == len(pairs):
            blocks.append(blk.digest()); blk = hashlib.sha256()
    rng = np.random.default_rng(20260927)
    ks = np.sort(rng.choice(len(pairs), size=min(nsample, len(pairs)), replace=False)).astype(np.uint32)
    t0 = time.time()
    L.ref_mpc_cons_pairs(ks.ctypes.data_as(u32p), len(ks))
    tB = time.time() - t0
    print("ConsPair x %d: %.1f s" % (len(ks), tB), flush=True)
    shas, snnz = [], []
    for k in ks:
        i, j = pairs[int(k)]
        nnz = L.ref_mpc_updated_nnz(int(k))
        off = np.empty(lens[i] + 1, np.uint32)
        val = np.empty(max(nnz, 1) * 2, np.uint32)
        L.ref_mpc_updated_sparse(int(k), off.ctypes.data_as(u32p), val.ctypes.data_as(u8p))
        shas.append(hashlib.sha256(off.tobytes() + val[:2 * nnz].tobytes()).digest()); snnz.append(nnz)
    d = {"n": np.int32(n), "length": np.int32(0), "seed": np.int32(0), "block": np.int32(BLOCK),
         "seqs_sha": np.array(hashlib.sha256("\n".join(seqs).encode()).hexdigest()),
         "ea_sha": np.array(hashlib.sha256(ea.tobytes()).hexdigest()),
         "ea_blocks": np.frombuffer(b"".join(eab), np.uint8).reshape(-1, 32),
         "nstages": np.int32(1), "digest0": np.array(whole.hexdigest()),
         "blocks0": np.frombuffer(b"".join(blocks), np.uint8).reshape(-1, 32), "nnz_total0": np.int64(tot),
         "sample_k": ks, "sample_nnz": np.array(snnz, np.uint32),
         "sample_sha1": np.frombuffer(b"".join(shas), np.uint8).reshape(-1, 32),
         "ref_seconds": np.array([tA, tB]), "ref_threads": np.int32(threads)}
    np.savez_compressed(os.path.join(HERE, "mpcbig_%s_sampled.npz" % name), **d)
    return name, n, tot, len(ks), [tA, tB]


def _clique_stage2(L, pairs, lens, clique, u32p, u8p):
    """STAGE 2 (after two relax iterations) of the pairs among `clique`, from the compiled reference, without relaxing the whole store:
    ConsPair of iteration 1 for every pair that touches the clique (what a second ConsPair of two clique members reads:
    conspairflat.cpp:49-89), the buffer swap of consflat.cpp:22, ConsPair of iteration 2 for the pairs among the clique.
    The store must be at stage 0. -> (ks, [sha256(offsets || values)], [nnz])."""
    import ctypes as C
    cs = set(int(c) for c in clique)
    touch = np.array([k for k, (i, j) in enumerate(pairs) if i in cs or j in cs], np.uint32)
    inner = np.array([k for k, (i, j) in enumerate(pairs) if i in cs and j in cs], np.uint32)
    L.ref_mpc_updated_nnz.restype = C.c_uint
    L.ref_mpc_cons_pairs(touch.ctypes.data_as(u32p), len(touch))
    L.ref_mpc_swap_stores()
    L.ref_mpc_cons_pairs(inner.ctypes.data_as(u32p), len(inner))
    shas, nnzs = [], []
    for k in inner:
        i, j = pairs[int(k)]
        nnz = L.ref_mpc_updated_nnz(int(k))
        off = np.empty(lens[i] + 1, np.uint32)
        val = np.empty(max(nnz, 1) * 2, np.uint32)
        L.ref_mpc_updated_sparse(int(k), off.ctypes.data_as(u32p), val.ctypes.data_as(u8p))
        shas.append(hashlib.sha256(off.tobytes() + val[:2 * nnz].tobytes()).digest())
        nnzs.append(nnz)
    return inner, shas, nnzs


def _mpc_stage2_worker(args):
    """Stage-2 pin at a size whose full relax takes days on the CPU: stage A of all pairs (the reference has to make its own stage 0),
    then _clique_stage2 for `m` seeded sequences. check=True (small n): the same pairs after two FULL ConsIter must give the same
    digests — the shortcut is checked against the thing it abbreviates (run in its own process: one MPCFlat per process)."""
    name, n, m, threads, check = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import time
    import _ref as R
    from muscle_amd.synth import make_family, read_fasta
    seqs = read_fasta(os.path.join(HERE, "rdrp_first1000.fa.gz"))[:n] if name.startswith("rdrp") else make_family(n, 60, seed=9)
    R.init_hmm(False, 0)
    L = R.lib()
    arr = (C.c_char_p * len(seqs))(*[s.encode() for s in seqs])
    if L.ref_mpc_begin(len(seqs), arr, threads) != 0:
        raise RuntimeError("ref_mpc_begin may only be called once per process")
    lens = [len(s) for s in seqs]
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    t0 = time.time()
    L.ref_mpc_calc_posteriors()
    tA = time.time() - t0
    print("stage A: %.1f s" % tA, flush=True)
    clique = np.sort(np.random.default_rng(20261001).choice(n, size=m, replace=False)).astype(np.uint32)
    if check:
        # two full iterations first, on a second look at the same store? No: one MPCFlat per process — the caller runs the full
        # version in another process and compares (tests/test_oracle_vs_ref.py)
        pass
    t0 = time.time()
    ks, shas, nnzs = _clique_stage2(L, pairs, lens, clique, R.u32p, R.u8p)
    tB = time.time() - t0
    print("stage 2 of %d pairs among %d sequences: %.1f s" % (len(ks), m, tB), flush=True)
    d = {"n": np.int32(n), "clique": clique, "stage2_k": ks, "stage2_nnz": np.array(nnzs, np.uint32),
         "stage2_sha": np.frombuffer(b"".join(shas), np.uint8).reshape(-1, 32),
         "seqs_sha": np.array(hashlib.sha256("\n".join(seqs).encode()).hexdigest()),
         "ref_seconds": np.array([tA, tB]), "ref_threads": np.int32(threads)}
    if not check:
        np.savez_compressed(os.path.join(HERE, "mpcbig_%s_stage2_clique.npz" % name), **d)
    return d


def gen_mpc_stage2(name, n, m, threads):
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        d = pool.map(_mpc_stage2_worker, [(name, n, m, threads, False)])[0]
        print(name, n, "clique", d["clique"].tolist(), len(d["stage2_k"]), "pairs", d["ref_seconds"].tolist(), flush=True)


def gen_mpc_sampled(name, n, nsample, threads):
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        print(pool.map(_mpc_sampled_worker, [(name, n, nsample, threads)])[0], flush=True)


def _mega_worker(name):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tempfile
    import _mega
    import _ref as R
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "in.mega")
        with open(path, "w") as f:
            f.write(_mega.mega_text(name))
        tables, stages, ea, probes = R.mpc_run_mega(path, iters=2, threads=0)
    d = {"seqs": np.array(tables["seqs"]), "alpha": tables["alpha"], "weight": tables["weight"],
         "lp": tables["lp"], "mx": tables["mx"], "profs": np.concatenate(tables["profs"]),
         "ea": ea, "nstages": np.int32(len(stages))}
    for k, v in probes.items():
        if k in ("F01", "B01"):  # M planes in full, digest of all five
            l0, l1 = len(tables["seqs"][0]), len(tables["seqs"][1])
            d[k + "_M"] = v.reshape(l0 + 1, l1 + 1, 5)[:, :, 0].copy()
            d[k + "_sha"] = np.array(sha(v))
        else:
            d[k] = v
    for s, st in enumerate(stages):
        d["digest%d" % s] = np.array(stage_digest(st))
        d["nnz%d" % s] = np.array([len(v) // 2 for _, v in st], np.uint32)
        d["off%d" % s] = np.concatenate([o for o, _ in st])
        d["val%d" % s] = np.concatenate([v for _, v in st]) if st else np.zeros(0, np.uint32)
    np.savez_compressed(os.path.join(HERE, "%s.npz" % name), **d)
    return name, len(tables["seqs"]), [int(sum(len(v) // 2 for _, v in st)) for st in stages]


MEGA_SETS = ["mega_bb11001", "mega_synth_6x40_s2", "mega_synth_3x25_s5_f3", "mega_synth_2x70_s7"]


def gen_mega():
    ctx = mp.get_context("spawn")
    for name in MEGA_SETS:  # one process per data set: Mega::FromFile / SetGlobalInputMS are once-per-process
        with ctx.Pool(1) as pool:
            print(pool.map(_mega_worker, [name])[0])


# ---- alignments of alignments: MPCFlat::BuildPost / AlignAlns and the pieces of PProg::AlignMSAsFlat, from the reference ----
BP_SETS = {"bp_n12_L70": (12, 70, 29), "bp_n9_L40": (9, 40, 3)}


def _bp_worker(args):
    """joins on the reference's own store after 2 ConsIter: the C1 x C2 matrix MPCFlat::BuildPost fills (buildpostflat.cpp:18-106),
    with and without weights, in both stored orientations (SMI_1 < SMI_2 and SMI_1 > SMI_2: :56-77 / :78-100), path and score of
    MPCFlat::AlignAlns (alnalnsflat.cpp:7-52); and for explicit pair lists the matrix of CalcPosteriorFlat3
    (buildposterior3flat.cpp:19-85), the path of CalcAlnFlat and the mean EA of GetPostPairsAlignedFlat (one thread)."""
    name, n, length, seed = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import _buildpost as BP
    import _ref as R
    from muscle_amd.synth import make_family
    seqs = make_family(n, length, seed=seed)
    R.init_hmm(False, 0)
    L = R.lib()
    arr = (C.c_char_p * n)(*[s.encode() for s in seqs])
    assert L.ref_mpc_begin(n, arr, 1) == 0
    L.ref_mpc_calc_posteriors()
    rng = np.random.default_rng(seed + 100)
    d = {"seqs": np.array(seqs)}
    u32p = C.POINTER(C.c_uint)

    def cargs(rows, idx):
        return len(rows), (C.c_char_p * len(rows))(*[r.encode() for r in rows]), np.ascontiguousarray(idx, np.uint32)

    # what `muscle -profseq msa -input2 query` logs (profseq.cpp:4-57: CalcPosterior for the pairs (row, query) only, NO
    # consistency, then MPCFlat::BuildPost + CalcAlnFlat): the path for an alignment of sequences 0..n-2 against sequence n-1,
    # from the stage-A store. (The reference's own command dies in GetGSIByLabel before it gets there.)
    ps_rows, _w = BP.random_msa(seqs, list(range(n - 1)), rng)
    n1, r1, i1 = cargs(ps_rows, list(range(n - 1)))
    n2, r2, i2 = cargs([seqs[n - 1]], [n - 1])
    buf = C.create_string_buffer(len(ps_rows[0]) + len(seqs[n - 1]) + 8)
    plen, score = C.c_uint(0), C.c_float(0)
    assert L.ref_mpc_align_alns(n1, r1, i1.ctypes.data_as(u32p), n2, r2, i2.ctypes.data_as(u32p), buf, C.byref(plen), C.byref(score)) == 0
    d["ps_rows"], d["ps_path"] = np.array(ps_rows), np.array(buf.raw[:plen.value].decode())
    for it in range(2):
        L.ref_mpc_cons_iter(it)

    joins = []
    order = [int(x) for x in rng.permutation(n)]
    for cut in (1, n // 2, n - 2):
        joins.append((order[:cut], order[cut:]))
    joins.append((sorted(order[:n // 2]), sorted(order[n // 2:])))
    joins.append(([n - 1], [0]))           # SMI_1 > SMI_2: the stored matrix read transposed
    joins.append(([1, 0], [n - 1, 2, 3]))  # both orientations inside one join
    nj = 0
    for grp1, grp2 in joins:
        rows1, C1 = BP.random_msa(seqs, grp1, rng)
        rows2, C2 = BP.random_msa(seqs, grp2, rng)
        n1, r1, i1 = cargs(rows1, grp1)
        n2, r2, i2 = cargs(rows2, grp2)
        post = np.zeros((C1, C2), np.float32)
        assert L.ref_mpc_build_post(n1, r1, i1.ctypes.data_as(u32p), n2, r2, i2.ctypes.data_as(u32p), None, 0, post.ctypes.data_as(R.f32p)) == 0
        # weights as m_Weights holds them: indexed by the row number inside each alignment (buildpostflat.cpp:42,52)
        w = rng.uniform(0.2, 1.8, n).astype(np.float32)
        postw = np.zeros((C1, C2), np.float32)
        assert L.ref_mpc_build_post(n1, r1, i1.ctypes.data_as(u32p), n2, r2, i2.ctypes.data_as(u32p), w.ctypes.data_as(R.f32p), n, postw.ctypes.data_as(R.f32p)) == 0
        buf = C.create_string_buffer(C1 + C2 + 8)
        plen, score = C.c_uint(0), C.c_float(0)
        assert L.ref_mpc_align_alns(n1, r1, i1.ctypes.data_as(u32p), n2, r2, i2.ctypes.data_as(u32p), buf, C.byref(plen), C.byref(score)) == 0
        k = "j%d_" % nj
        d[k + "idx1"], d[k + "idx2"] = np.array(grp1, np.uint32), np.array(grp2, np.uint32)
        d[k + "rows1"], d[k + "rows2"] = np.array(rows1), np.array(rows2)
        d[k + "post"], d[k + "postw"], d[k + "w"] = post, postw, w
        d[k + "path"], d[k + "score"] = np.array(buf.raw[:plen.value].decode()), np.float32(score.value)
        nj += 1
    d["njoins"] = np.int32(nj)
    # explicit pair lists (PProg joins): all pairs, and a subset in scrambled order
    nm = 0
    for grp1, grp2, keep in ((order[:3], order[3:8], None), (order[5:], order[:4], lambda a, b: (a * 7 + b) % 5 != 2)):
        rows1, C1 = BP.random_msa(seqs, grp1, rng)
        rows2, C2 = BP.random_msa(seqs, grp2, rng)
        pairs = [(a, b) for a in range(len(grp1)) for b in range(len(grp2)) if keep is None or keep(a, b)]
        if keep is not None:
            pairs = [pairs[int(q)] for q in rng.permutation(len(pairs))]
        s1 = np.array([a for a, b in pairs], np.uint32)
        s2 = np.array([b for a, b in pairs], np.uint32)
        n1, r1, i1 = cargs(rows1, grp1)
        n2, r2, i2 = cargs(rows2, grp2)
        post = np.zeros((C1, C2), np.float32)
        buf = C.create_string_buffer(C1 + C2 + 8)
        plen, ea = C.c_uint(0), C.c_float(0)
        assert L.ref_align_msas(n1, r1, i1.ctypes.data_as(u32p), n2, r2, i2.ctypes.data_as(u32p), len(pairs), s1.ctypes.data_as(u32p),
                                s2.ctypes.data_as(u32p), post.ctypes.data_as(R.f32p), buf, C.byref(plen), C.byref(ea)) == 0
        k = "m%d_" % nm
        d[k + "idx1"], d[k + "idx2"] = np.array(grp1, np.uint32), np.array(grp2, np.uint32)
        d[k + "rows1"], d[k + "rows2"] = np.array(rows1), np.array(rows2)
        d[k + "row1"], d[k + "row2"] = s1, s2  # per pair: row of MSA1 / row of MSA2
        d[k + "post"], d[k + "path"], d[k + "ea_avg"] = post, np.array(buf.raw[:plen.value].decode()), np.float32(ea.value)
        nm += 1
    d["nmsas"] = np.int32(nm)
    np.savez_compressed(os.path.join(HERE, "%s.npz" % name), **d)
    return name, nj, nm


def gen_bp():
    ctx = mp.get_context("spawn")
    for name, (n, length, seed) in BP_SETS.items():
        with ctx.Pool(1) as pool:
            print(pool.map(_bp_worker, [(name, n, length, seed)])[0], flush=True)


# ---- AlignPairFlat (alignpairflat.cpp:3-27): path, EA and FromPost matrix of single pairs, from the reference ----------
def _ap_worker(args):
    name, seqs, pairs = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import _ref as R
    R.init_hmm(False, 0)
    L = R.lib()
    n = len(seqs)
    arr = (C.c_char_p * n)(*[s.encode() for s in seqs])
    assert L.ref_mpc_begin(n, arr, 1) == 0
    u32p = C.POINTER(C.c_uint)
    d = {"seqs": np.array(seqs), "pairs": np.array(pairs, np.uint32)}
    for q, (i, j) in enumerate(pairs):
        LX, LY = len(seqs[i]), len(seqs[j])
        buf = C.create_string_buffer(LX + LY + 8)
        plen, ea, nnz = C.c_uint(0), C.c_float(0), C.c_uint(0)
        off = np.empty(LX + 1, np.uint32)
        val = np.empty(max(LX * LY, 1) * 2, np.uint32)
        assert L.ref_align_pair(i, j, buf, C.byref(plen), C.byref(ea), C.byref(nnz), off.ctypes.data_as(u32p),
                                val.ctypes.data_as(R.u8p), LX * LY) == 0
        d["p%d_path" % q], d["p%d_ea" % q] = np.array(buf.raw[:plen.value].decode()), np.float32(ea.value)
        d["p%d_off" % q], d["p%d_val" % q] = off, val[:2 * nnz.value].copy()
    np.savez_compressed(os.path.join(HERE, "%s.npz" % name), **d)
    return name, len(pairs)


def gen_ap():
    from muscle_amd.synth import make_family
    fam = make_family(6, 90, seed=17)
    seqs = fam + ["M", "MKVLA", make_family(1, 300, seed=9)[0], fam[2], "ACDEFGHIKLMNPQRSTVWY" * 3, "WWWWWWWW"]
    pairs = [(0, 1), (1, 0), (2, 5), (6, 7), (7, 8), (8, 0), (3, 9), (9, 3), (10, 4), (11, 6), (8, 10), (4, 4 + 1)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        print(pool.map(_ap_worker, [("ap_ragged", seqs, pairs)])[0], flush=True)


if __name__ == "__main__":
    if sys.argv[1:] == ["ap"]:
        gen_ap()
        sys.exit(0)
    if sys.argv[1:] == ["bp"]:
        gen_bp()
        sys.exit(0)
    if sys.argv[1:] == ["mega"]:
        gen_mega()
        sys.exit(0)
    if sys.argv[1:2] == ["big"]:  # python make_golden.py big <threads> <name> ...: digest-only BASELINE-size sets (CPU-hours)
        gen_mpc_big(sys.argv[3:], int(sys.argv[2]))
        sys.exit(0)
    if sys.argv[1:2] == ["big-stage2"]:  # python make_golden.py big-stage2 <threads> <n> [m]: rdrp prefix, stage 2 of the pairs among m seeded sequences
        gen_mpc_stage2("rdrp%d" % int(sys.argv[3]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 6, int(sys.argv[2]))
        sys.exit(0)
    if sys.argv[1:2] == ["big-sampled"]:  # python make_golden.py big-sampled <threads> <n> [nsample]: rdrp prefix, stage A of all pairs + ConsPair of a sample
        gen_mpc_sampled("rdrp%d" % int(sys.argv[3]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 2048, int(sys.argv[2]))
        sys.exit(0)
    if sys.argv[1:2] == ["mpc"]:  # python make_golden.py mpc <name> ...: only these whole-stage sets
        gen_mpc(set(sys.argv[2:]))
        sys.exit(0)
    gen_hmm()
    gen_pairs_small()
    gen_mpc()
    gen_mega()
    print("golden fixtures written to", HERE)
