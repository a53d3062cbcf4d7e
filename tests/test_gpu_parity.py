"""Parity tests proper (-m gpu): the HIP path through the C ABI vs the CPU oracle / golden fixtures
on the same inputs. Bit-exact: integer structure (offsets, columns) AND float posteriors/EA (the
north star allows 1e-4 on floats; we hold 0 ulp because the final MSA depends on exact values)."""
import os

import numpy as np
import pytest

import _golden as G
import _parity as P
from muscle_amd._lib import MpcGpu, MpcGpuError
from muscle_amd.synth import make_family

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", G.MPC_SETS_GPU)
def test_golden_sets(name):
    g = G.mpc(name)
    stages, ea = P.run_lib(g["seqs"])
    assert np.array_equal(P.bits(ea), P.bits(g["ea"]))
    for s in range(g["nstages"]):
        assert np.array_equal(np.array([len(v) // 2 for _, v in stages[s]], np.uint32), g["nnz"][s])
        assert G.stage_digest(stages[s]) == g["digest"][s], "stage %d" % s


@pytest.mark.parametrize("name", G.MEGA_SETS)
def test_mega_golden_sets(name):
    """Structure-profile emissions (mpcgpu_set_mega; calcpost.cpp:14-22 -> fwdflat_mega.cpp / bwdflat_mega.cpp):
    the reference's parsed tables in, the reference's own stage outputs expected."""
    m = G.mega(name)
    stages, ea = P.run_lib(m["seqs"], mega=m)
    assert np.array_equal(P.bits(ea), P.bits(m["ea"]))
    for s in range(m["nstages"]):
        assert G.stage_digest(stages[s]) == m["digest"][s], "stage %d" % s
        for (o1, v1), (o2, v2) in zip(stages[s], m["stage"][s]):
            assert np.array_equal(o1, o2) and np.array_equal(v1, v2)


@pytest.mark.parametrize("nfeat", [8, 5, 1])
def test_mega_vs_oracle_ragged(nfeat):
    """random tables; lengths 1 .. 700 (H = 1 .. 11 rows per lane), fewer than 8 features (unused ones read the 0 entry)"""
    seqs = ["M", "MKVLA", make_family(1, 700, seed=2)[0], make_family(1, 420, seed=3)[0], make_family(1, 64, seed=4)[0],
            make_family(1, 65, seed=5)[0], make_family(1, 129, seed=6)[0]]
    mega = P.random_mega(seqs, seed=20 + nfeat, nfeat=nfeat)
    P.assert_same(P.run_lib(seqs, mega=mega), P.run_oracle(seqs, mega=mega), "mega ragged F=%d" % nfeat)


def test_mega_then_letters_on_one_context():
    """set_seqs drops the profiles: the same context must go back to letter emissions"""
    s, t, m, i, thr = G.hmm_tables()
    seqs = make_family(4, 70, seed=31)
    mega = P.random_mega(seqs, seed=5)
    g = MpcGpu(0)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs(seqs)
    g.set_mega(mega["alpha"], mega["weight"], mega["lp"], mega["mx"], mega["profs"])
    g.calc_posteriors()
    ea_mega = g.get_ea().copy()
    g.set_seqs(seqs)
    g.calc_posteriors()
    ea_plain = g.get_ea().copy()
    g.close()
    assert np.array_equal(P.bits(ea_mega), P.bits(P.run_oracle(seqs, iters=0, mega=mega)[1]))
    assert np.array_equal(P.bits(ea_plain), P.bits(P.run_oracle(seqs, iters=0)[1]))


def test_nucleotide_tables():
    rng = np.random.default_rng(3)
    seqs = ["".join(rng.choice(list("ACGU" if k % 2 else "ACGT"), size=int(rng.integers(20, 120)))) for k in range(7)]
    P.assert_same(P.run_lib(seqs, hmm_name="hmm_nucleo"), P.run_oracle(seqs, hmm_name="hmm_nucleo"), "nucleo")


def test_wildcards_and_lowercase():
    seqs = ["XXBZACDEF", "AXCBJOU", "acdefGHIKL", "MKVLAXXXX", "ACDEFGHIKLMNPQRSTVWYBZX"]
    P.assert_same(P.run_lib(seqs), P.run_oracle(seqs), "wildcards")


def test_ragged_and_extremes():
    long1024 = (make_family(1, 1100, seed=5)[0] * 2)[:1024]  # exactly the longest row sequence this build takes
    seqs = ["M", "W", "MKVLA", make_family(1, 900, seed=2)[0], make_family(1, 1000, seed=3)[0][:1024], long1024,
            "ACDEFGHIKLMNPQRSTVWY" * 10, make_family(1, 333, seed=4)[0]]
    P.assert_same(P.run_lib(seqs), P.run_oracle(seqs, threads=0), "ragged")


def test_eight_rows_per_lane():
    """449..512 residues: fb_kernel<8>, the instantiation held to 128 VGPRs by its launch bounds (a few spilled dwords)"""
    seqs = make_family(6, 480, seed=71) + [make_family(1, 512, seed=72)[0][:512], make_family(1, 449, seed=73)[0][:449]]
    P.assert_same(P.run_lib(seqs), P.run_oracle(seqs, threads=0), "8 rows per lane")


def test_long_rows_row_blocks():
    """X longer than 64*16 rows: the row-block kernel (fb_kernel<7, MEGA, LONG>: 448-row blocks chained through
    the line buffers). 1024 is the last single-block length, 1025 the first with blocks; with 2049 columns the row-list
    post kernel needs more than 64 KB of LDS; the relax of such records takes the gather fallback."""
    seqs = [make_family(1, 1500, seed=41)[0][:1500], (make_family(1, 1100, seed=42)[0] * 2)[:1025],
            (make_family(1, 1100, seed=43)[0] * 2)[:1024], make_family(1, 333, seed=44)[0],
            (make_family(1, 2300, seed=45)[0] * 2)[:2049]]
    P.assert_same(P.run_lib(seqs), P.run_oracle(seqs, threads=0), "row blocks")


def test_long_rows_row_lists_and_mega():
    """row-block pairs through the default post kernel (16-bit column keys), then with structure profiles"""
    seqs = [make_family(1, 1400, seed=51)[0], make_family(1, 1030, seed=52)[0], make_family(1, 200, seed=53)[0]]
    P.assert_same(P.run_lib(seqs), P.run_oracle(seqs, threads=0), "row blocks, row lists")
    mega = P.random_mega(seqs, seed=13)
    P.assert_same(P.run_lib(seqs, mega=mega), P.run_oracle(seqs, mega=mega, threads=0), "row blocks, mega")


def test_long_related_sequences():
    """two RELATED sequences of ~2600 residues (plus a short one): row blocks in both directions of the pair list, more stored
    cells per pair than the row-list post kernel's LDS list holds (its global scratch), EA rows over 2600 columns"""
    fam = make_family(2, 2600, seed=61)
    seqs = [fam[0], fam[1], make_family(1, 120, seed=62)[0]]
    P.assert_same(P.run_lib(seqs), P.run_oracle(seqs, threads=0), "long related sequences")
    got = None
    os.environ["MPCGPU_POST"] = "sort"  # the general post kernel on the same input
    try:
        got = P.run_lib(seqs)
    finally:
        del os.environ["MPCGPU_POST"]
    P.assert_same(got, P.run_oracle(seqs, threads=0), "long related sequences, general post kernel")
    os.environ["MPCGPU_FB_LONG_H"] = "4"  # the 4-rows-per-lane row-block kernel (chosen by itself only when thousands of long pairs are resident)
    try:
        got = P.run_lib(seqs)
    finally:
        del os.environ["MPCGPU_FB_LONG_H"]
    P.assert_same(got, P.run_oracle(seqs, threads=0), "long related sequences, 4 rows per lane")


def test_very_long_row_sequence():
    """a 9048-residue row sequence against short ones: 21 row blocks in the fb kernel, 16-bit-column candidate keys,
    and (longer than 8191) the gather relax instead of the LDS tiles"""
    fam = make_family(2, 50, seed=3)
    big = make_family(1, 9000, seed=4)[0]
    big = big[:4000] + fam[0] + big[4000:]  # related to the short ones somewhere in the middle
    seqs = [big, fam[0], fam[1]]
    P.assert_same(P.run_lib(seqs), P.run_oracle(seqs), "9048-long row sequence")


def test_identical_sequences_saturate():
    s = make_family(1, 120, seed=8)[0]
    seqs = [s, s, s[:100], s[10:]]
    P.assert_same(P.run_lib(seqs), P.run_oracle(seqs), "identical")


def test_both_expf_variants():
    seqs = make_family(6, 90, seed=12)
    a = P.run_lib(seqs, expf_variant=0)
    b = P.run_lib(seqs, expf_variant=1)
    o = P.run_oracle(seqs)
    # the host's libm resolves to one of the two; that one must be bit-identical, the other within 1 ulp
    import _oracle as O
    fma = O.lib().orc_host_expf_uses_fma()
    P.assert_same(b if fma else a, o, "host expf variant")
    for (o1, v1), (o2, v2) in zip(a[0][0], b[0][0]):
        assert np.array_equal(o1, o2)
        assert np.max(np.abs(v1[0::2].astype(np.int64) - v2[0::2].astype(np.int64))) <= 1


def test_config2_shape_256x300_vs_oracle_sample():
    """BASELINE config 2 shape (256 seqs L~300) is too slow for the scalar oracle in full; check a
    64-sequence sub-family in full and the size-independent properties on the 256 set."""
    fam = make_family(256, 300, seed=1)
    sub = fam[:40]
    P.assert_same(P.run_lib(sub), P.run_oracle(sub), "40x300")
    stages, ea = P.run_lib(fam)
    n = len(fam)
    assert np.all((ea >= 0) & (ea <= 1.0))
    for s in range(3):
        for k, (off, val) in enumerate(stages[s]):
            p = val[0::2].view(np.float32)
            assert off[0] == 0 and off[-1] == len(p) and np.all(np.diff(off.astype(np.int64)) >= 0)
            assert np.all(p >= 0) and np.all(p <= 1.0 + 1e-6)
    # the pattern is frozen by relax (mysparsemx.cpp:97-112): offsets and columns identical across stages
    for k in range(len(stages[0])):
        assert np.array_equal(stages[0][k][0], stages[2][k][0])
        assert np.array_equal(stages[0][k][1][1::2], stages[2][k][1][1::2])


def test_sharded_equals_whole():
    """Pair-sharded stage A + import of the shards (the multi-GPU path, here on one device through
    a host staging copy) must give the same store as the unsharded run."""
    seqs = make_family(9, 110, seed=21)
    whole = P.run_lib(seqs)
    s, t, m, i, thr = G.hmm_tables()
    np_ = 9 * 8 // 2
    cuts = [0, 7, 20, np_]
    ctxs, blobs = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        g = MpcGpu(0)
        g.set_hmm(s, t, m, i, thr)
        g.set_seqs(seqs)
        g.calc_posteriors(a, b)
        ctxs.append(g)
        blobs.append(g.shard_info())
    # caller-owned device memory for the "gathered" shards: straight from the HIP runtime (no torch:
    # importing torch AFTER libmpcgpu has loaded the system HIP runtime breaks torch's device init)
    import ctypes as C
    hip = None
    for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = C.CDLL(name)
            break
        except OSError:
            continue
    assert hip is not None
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    total = sum(b for b, _ in blobs)
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(total)) == 0
    off = 0
    for c, (nbytes, _) in zip(ctxs, blobs):
        c.shard_export(dptr.value + off)
        off += nbytes
    g = ctxs[0]
    g.store_import(cuts[:-1], cuts[1:], [b for b, _ in blobs], dptr.value)
    stages = [g.get_sparse_range()]
    ea = g.get_ea()
    for _ in range(2):
        # shard the relax too, in two halves, then commit once
        g.cons_iter(0, 11)
        g.cons_iter(11, np_)
        g.cons_commit()
        stages.append(g.get_sparse_range())
    P.assert_same((stages, ea), whole, "sharded")
    for c in ctxs:
        c.close()
    hip.hipFree(dptr)


def test_two_rank_exchange_through_torch_tensors():
    """bench.py's N>1 code (muscle_amd.mpcflat.run_stage + TorchExchange) with two ranks as two threads on this
    one GPU and an in-process stand-in for the collectives: shards and values travel through torch CUDA tensors
    (allocator/pointer interop with the library), store_import reads them, both ranks end bit-identical to a
    single context. Fresh process: torch has to be imported before the library (tests/_torch_exchange_check.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PYTHONPATH=os.path.dirname(here) + os.pathsep + here)
    env.pop("TEC_DEVICE", None)
    env.pop("TEC_LIB", None)
    r = subprocess.run([sys.executable, "-u", os.path.join(here, "_torch_exchange_check.py")], env=env, cwd=here,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240, text=True)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "OK two-rank exchange" in r.stdout, r.stdout[-3000:]


def test_back_to_back_iterations_without_sync():
    """The drop-in (hostcxx/mpcflat_gpu.cpp) and bench.py queue both relax iterations without reading
    anything back in between; the result must not depend on host-side synchronisation (regression:
    the tile list of the second iteration was uploaded from a temporary that had gone out of scope)."""
    for n, L, seed in ((8, 60, 31), (14, 90, 32), (23, 40, 33)):
        seqs = make_family(n, L, seed=seed)
        s, t, m, i, thr = G.hmm_tables()
        g = MpcGpu(0)
        g.set_hmm(s, t, m, i, thr)
        g.set_seqs(seqs)
        g.calc_posteriors()
        g.build_store()
        for _ in range(2):
            g.cons_iter()
            g.cons_commit()
        got = g.get_sparse_range()
        g.close()
        want = P.run_oracle(seqs)[0][2]
        for k, ((o1, v1), (o2, v2)) in enumerate(zip(got, want)):
            assert np.array_equal(o1, o2) and np.array_equal(v1, v2), "n=%d pair %d" % (n, k)


def test_relax_gather_equals_tiled():
    """Both relax kernels (LDS-tiled over variable-size records = default in its geometries, one-cell-per-thread gather
    fallback) are device code with the reference's accumulation order: identical bits."""
    import os
    seqs = make_family(21, 120, seed=41)
    a = P.run_lib(seqs)
    os.environ["MPCGPU_RELAX"] = "gather"
    try:
        b = P.run_lib(seqs)
    finally:
        del os.environ["MPCGPU_RELAX"]
    P.assert_same(a, b, "gather vs tiled")
    P.assert_same(a, P.run_oracle(seqs), "tiled vs oracle")
    # relax_band_kernel (the default above) in forced tile shapes and staging modes, with the compiler's merge instead of the
    # hand-scheduled one (the shipped asm is pinned only here, on hardware: the emulator runs the C++ statement), and in its
    # four-workgroups-per-CU geometry; relax_var_kernel (whole-record tiles) in its geometries
    variants = [{"MPCGPU_RELAX_SHAPE": "8,8"}, {"MPCGPU_RELAX_SHAPE": "4,2,12"}, {"MPCGPU_RELAX_SHAPE": "1,1"},
                {"MPCGPU_RELAX_SHAPE": "8,8", "MPCGPU_RELAX_SLOTS": "2"}, {"MPCGPU_RELAX_LDS_KB": "24"},
                {"MPCGPU_RELAX_MERGE": "cxx"}, {"MPCGPU_RELAX_WG": "512"}, {"MPCGPU_RELAX_ORDER": "pairs"}, {"MPCGPU_RELAX_ORDER": "pairs", "MPCGPU_RELAX_FORM": "walk"}, {"MPCGPU_RELAX_ORDER": "1"}, {"MPCGPU_RELAX_ORDER": "5", "MPCGPU_RELAX_FORM": "walk"},
                # the default above looks the Y rows up in window records (narrow rows); the two-list walk on block records, and both merges of it
                {"MPCGPU_RELAX_FORM": "walk"}, {"MPCGPU_RELAX_FORM": "walk", "MPCGPU_RELAX_MERGE": "cxx"}, {"MPCGPU_RELAX_FORM": "walk", "MPCGPU_RELAX_SHAPE": "4,2,12"}]
    variants += [{"MPCGPU_RELAX_TILES": "pairs", "MPCGPU_RELAX_WG": geo, "MPCGPU_RELAX_NBUF": nbuf}
                 for geo, nbuf in (("2048", "1"), ("1024", "2"), ("1024", "1"), ("512", "1"), ("768", "1"))]
    for env in variants:
        os.environ.update(env)
        try:
            info = {}
            r = P.run_lib(seqs, info=info)
        finally:
            for k in env:
                del os.environ[k]
        P.assert_same(a, r, "relax variant %s" % env)
        assert ("relax_var_kernel" if "MPCGPU_RELAX_TILES" in env else "relax_band_kernel") in info["relax_info"], (env, info["relax_info"])


def test_random_ragged_stores_against_the_oracle():
    """seeded random inputs through the default path (band tiles, window records where they pay, row-block cell order): mixes of
    families of different lengths, unrelated sequences, fragments of 1..5 residues — every stage bit-identical to the oracle.
    MPCGPU_TEST_FUZZ_CASES widens the sweep (default 40 cases: 2 s on the GPU box)."""
    import random
    cases = int(os.environ.get("MPCGPU_TEST_FUZZ_CASES", "40"))
    for case in range(cases):
        rng = random.Random(1000 + case)
        seqs = []
        for _ in range(rng.randint(1, 3)):
            seqs += make_family(rng.randint(2, 14), rng.choice([12, 30, 60, 90, 150, 240]), seed=rng.randint(1, 10 ** 6),
                                p_del=rng.choice([0.03, 0.1]), p_ins=rng.choice([0.03, 0.1]), p_sub=rng.choice([0.2, 0.5]))
        for _ in range(rng.randint(0, 3)):
            seqs.append(make_family(1, rng.randint(1, 5), seed=rng.randint(1, 10 ** 6))[0])
        if rng.random() < 0.5:  # fragments of a family member: one-sided overlaps, wide rows against the full-length ones
            t = seqs[0]
            seqs += [t[: max(1, len(t) // 3)], t[len(t) // 2:]]
        rng.shuffle(seqs)
        info = {}
        got = P.run_lib(seqs, info=info)
        P.assert_same(got, P.run_oracle(seqs), "fuzz case %d (%d sequences, lengths %s; %s)" % (case, len(seqs), [len(t) for t in seqs], info.get("relax_info", "")[-80:]))


def test_fuzz_generator_slice_single_context_joins_and_groups():
    """A slice of diag/fuzz_parity.py's seeded generator (tiny / long / low-complexity / identical / fragment / nucleotide / byte
    sequences, rdrp picks, mixtures; 1..3 relax iterations): every stage against the oracle, random joins (BuildPost + CalcAlnFlat)
    against the restatement, and the same cases as a group of 2..8 contexts — every rank, every stage. The long runs are in
    profiles/r14c, r14d (2860 cases); MPCGPU_TEST_FUZZ_SEEDS widens this one (default 48 seeds: ~25 s on the GPU box)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diag"))
    import fuzz_parity as F
    joins = groups = 0
    nseeds = int(os.environ.get("MPCGPU_TEST_FUZZ_SEEDS", "48"))
    for seed in range(3000, 3000 + nseeds):
        what, seqs, hmm, iters = F.make_case(seed)
        got = P.run_lib(seqs, iters=iters, hmm_name=hmm)
        want = P.run_oracle(seqs, iters=iters, hmm_name=hmm)
        P.assert_same(got, want, "seed %d: %s" % (seed, what))
        joins += F.check_joins(seqs, hmm, seed)
        if seed % 3 == 0:
            groups += F.check_group(seqs, hmm, iters, want, seed)
    assert nseeds < 24 or (joins > 0 and groups > 0)


def test_relax_cell_order_on_long_row_bands(monkeypatch):
    """the row-block cell order where a tile's band is the whole sequence (few pairs: 500 rows, the order's tables take 72 KB of the
    staging area) and where the tables do not fit (800 rows: the kernel falls back to the pair order by itself) — against the
    oracle, and the orders against each other"""
    for n, length in ((12, 500), (6, 800)):
        seqs = make_family(n, length, seed=7)
        want = P.run_oracle(seqs)
        for order in (None, "1", "pairs"):
            if order:
                monkeypatch.setenv("MPCGPU_RELAX_ORDER", order)
            info = {}
            got = P.run_lib(seqs, info=info)
            if order:
                monkeypatch.delenv("MPCGPU_RELAX_ORDER")
            assert "relax_band_kernel" in info["relax_info"], info["relax_info"]
            P.assert_same(got, want, "%d x %d, order %s" % (n, length, order))


def test_relax_window_rows_wider_than_the_span_field(monkeypatch):
    """the direct-index merge's escape (a window descriptor's 5-bit span field is 31: the span comes from the next row's offset) in
    its hand-scheduled form: ragged unrelated sequences — a 3-residue row against 75 columns — with window records forced"""
    seqs = make_family(7, 75, seed=11) + make_family(3, 18, seed=5) + [make_family(1, 70, seed=9)[0], "MKV"]
    want = P.run_oracle(seqs)
    assert max(int(v[1::2][o[i + 1] - 1]) - int(v[1::2][o[i]]) + 1 for o, v in want[0][0] for i in range(len(o) - 1) if o[i + 1] > o[i]) > 31
    monkeypatch.setenv("MPCGPU_RELAX_WIN_PCT", "100000")
    for extra in ({}, {"MPCGPU_RELAX_ORDER": "1"}, {"MPCGPU_RELAX_SHAPE": "4,2", "MPCGPU_RELAX_SLOTS": "2"}, {"MPCGPU_RELAX_MERGE": "cxx"}):
        for k, v in extra.items():
            monkeypatch.setenv(k, v)
        info = {}
        got = P.run_lib(seqs, info=info)
        for k in extra:
            monkeypatch.delenv(k)
        assert "MpcRbWin" in info["relax_info"], info["relax_info"]
        P.assert_same(got, want, "window escape %s" % extra)


@pytest.mark.parametrize("kernel", ["by size", "one wave", "waves", "lds rows"])
def test_calc_aln_paths(kernel, monkeypatch):
    """CalcAlnFlat + TraceBackFlat on the device: integer traceback bit-for-bit (path string) and
    score bits, vs the golden paths of the compiled reference (pairs_small) and vs the oracle on
    dense MSA-sized matrices with many exact ties."""
    # calc_aln_wave_kernel (<= 512 columns) / calc_aln_quad_kernel (<= 4096 columns) / calc_aln_kernel; a kernel that cannot take
    # a matrix leaves it to the next one
    monkeypatch.setenv("MPCGPU_ALN_KERNEL", str(["by size", "one wave", "waves", "lds rows"].index(kernel)))
    import _oracle as O
    g = MpcGpu(0)
    z = G.load("pairs_small")
    for k in range(int(z["n"])):
        path, sc = g.calc_aln(z["post%d" % k])
        assert path == str(z["path%d" % k])
        assert P.bits(sc) == P.bits(z["calcaln_score%d" % k])
    rng = np.random.default_rng(11)
    # (255 .. 511 columns: both sides of every columns-per-lane class of the one-wave kernel, 4 .. 8)
    for LX, LY in ((1, 1), (1, 9), (7, 1), (150, 170), (620, 580), (300, 2500), (1010, 990), (2000, 4000), (40, 4500),
                   (90, 255), (90, 256), (310, 319), (90, 320), (400, 383), (90, 384), (450, 447), (90, 448), (500, 511)):
        M = ((rng.random((LX, LY)) < 0.02) * rng.random((LX, LY)) * 3).astype(np.float32)
        for Q in (M, np.round(M * 2) / 2):
            path, sc = g.calc_aln(Q.astype(np.float32))
            sc0, path0 = O.calc_aln(Q.astype(np.float32))
            assert path == path0 and P.bits(sc) == P.bits(sc0), (LX, LY)
    g.close()


def test_align_alns_vs_restatement():
    """mpcgpu_align_alns (device BuildPost by sorted in-order reduction + CalcAlnFlat + traceback) vs the
    numpy restatement of buildpostflat.cpp and the oracle's CalcAlnFlat on random gapped alignments of
    a 12-sequence family after two relax iterations: identical path strings and score bits."""
    import _buildpost as BP
    import _oracle as O
    rng = np.random.default_rng(4)
    n = 12
    seqs = make_family(n, 70, seed=29)
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs(seqs)
    g.calc_posteriors()
    g.build_store()
    for _ in range(2):
        g.cons_iter()
        g.cons_commit()
    stage = g.get_sparse_range()
    pidx = {p: k for k, p in enumerate((a, b) for a in range(n) for b in range(a + 1, n))}
    order = list(rng.permutation(n))
    for cut in (1, 5, 6, 11):
        grp1, grp2 = [int(x) for x in order[:cut]], [int(x) for x in order[cut:]]
        rows1, C1 = BP.random_msa(seqs, grp1, rng)
        rows2, C2 = BP.random_msa(seqs, grp2, rng)
        m1 = [BP.pos_to_col(r) for r in rows1]
        m2 = [BP.pos_to_col(r) for r in rows2]
        post = BP.build_post(stage, pidx, grp1, grp2, m1, m2, C1, C2)
        sc0, path0 = O.calc_aln(post)
        path, sc = g.align_alns(grp1, grp2, m1, m2, C1, C2)
        assert path == path0 and P.bits(sc) == P.bits(sc0), cut
        # sequence weights (buildpostflat.cpp:41,52,74: every contribution is (w1*w2)*P)
        w1 = rng.uniform(0.2, 1.8, len(grp1)).astype(np.float32)
        w2 = rng.uniform(0.2, 1.8, len(grp2)).astype(np.float32)
        sc0w, path0w = O.calc_aln(BP.build_post(stage, pidx, grp1, grp2, m1, m2, C1, C2, w1, w2))
        pathw, scw = g.align_alns(grp1, grp2, m1, m2, C1, C2, w1, w2)
        assert pathw == path0w and P.bits(scw) == P.bits(sc0w), ("weighted", cut)
    # timing off (mpcgpu_timers_enable: what the drop-in runs with): same results, nothing measured
    g.timers_reset()
    g.timers_enable(False)
    path, sc = g.align_alns(grp1, grp2, m1, m2, C1, C2)
    assert path == path0 and P.bits(sc) == P.bits(sc0)
    assert all(v == (0.0, 0) for v in g.timers_get().values())
    g.timers_enable(True)
    g.close()


def test_align_msas_vs_restatement():
    """mpcgpu_align_msas (PProg join: stage A on an explicit pair list in both index orders,
    CalcPosteriorFlat3 in pair-list order, CalcAlnFlat + traceback) vs the oracle per pair and the numpy
    restatement of buildposterior3flat.cpp:19-85: identical path, score bits and per-pair EA bits."""
    import _buildpost as BP
    import _oracle as O
    rng = np.random.default_rng(19)
    seqs = make_family(14, 90, seed=37)
    s, t, m, i, thr = G.hmm_tables()
    h = O.make_hmm(s, t, m, i)
    g = MpcGpu(0)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs_registry(seqs)
    grp1, grp2 = [9, 0, 3, 12, 5], [7, 1, 2, 6, 4, 13, 8, 10, 11]
    rows1, C1 = BP.random_msa(seqs, grp1, rng)
    rows2, C2 = BP.random_msa(seqs, grp2, rng)
    pairs = [(a, b) for a in range(len(grp1)) for b in range(len(grp2)) if (a * 7 + b) % 5 != 2]
    seq1 = [grp1[a] for a, b in pairs]
    seq2 = [grp2[b] for a, b in pairs]
    m1 = [BP.pos_to_col(rows1[a]) for a, b in pairs]
    m2 = [BP.pos_to_col(rows2[b]) for a, b in pairs]
    post = np.zeros((C1, C2), np.float32)
    ea_want = []
    for q, (X, Y) in enumerate(zip(seq1, seq2)):
        x, y = seqs[X].encode(), seqs[Y].encode()
        Pd = O.post(O.fwd(h, x, y), O.bwd(h, x, y), len(x), len(y))
        ea_want.append(np.float32(O.aln_score(Pd)) / np.float32(min(len(x), len(y))))
        off, val = O.sparse_from_post(Pd)
        p, col = val[0::2].view(np.float32), val[1::2]
        for r in range(len(off) - 1):
            for k in range(off[r], off[r + 1]):
                post[m1[q][r], m2[q][col[k]]] += p[k]  # buildposterior3flat.cpp:81
    sc0, path0 = O.calc_aln(post)
    path, sc, ea = g.align_msas(seq1, seq2, m1, m2, C1, C2)
    assert path == path0 and P.bits(sc) == P.bits(sc0)
    assert np.array_equal(P.bits(ea), P.bits(np.array(ea_want, np.float32)))
    g.close()


def test_errors_are_loud():
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0)
    with pytest.raises(MpcGpuError):
        g.set_seqs(["ACD", "ACD"])  # no HMM yet
    g.set_hmm(s, t, m, i, thr)
    with pytest.raises(MpcGpuError):
        g.set_seqs(["ACD"])  # need >= 2
    with pytest.raises(MpcGpuError):
        g.set_seqs(["ACD", ""])
    g.set_seqs(["ACD", "ACE", "ACF"])
    with pytest.raises(MpcGpuError):
        g.cons_iter()  # no store yet
    with pytest.raises(MpcGpuError):
        g.calc_posteriors(2, 1)
    g.close()


@pytest.mark.parametrize("name", ["n256_L300", "n1000_L400", "rdrp256", "rdrp384"])
def test_baseline_configs_vs_reference_digests(name):
    """BASELINE configs 2 (256 x L~300) and 3 (1000 x L~400, the benchmarked workload) in full through the default
    kernels: EA bits and all three stage snapshots (after CalcPosteriors, after each ConsIter) against digests
    generated by the compiled reference (tests/golden/mpcbig_*.npz; mpcflat.cpp:313,328, mysparsemx.cpp:87-113).
    rdrp256: real data (first 256 records of the reference's test_data/rdrp), ~7 stored cells per row: records of tens of KB,
    band tiles of <= 4x2 pairs with the two-list walk, as the 1000-record bench run uses. rdrp384: the first 384 records (73 536
    pairs, 202.8 M stored entries; 2.5 h of the compiled reference on 6 threads): more than 256 sequences put more than one tile
    row of X groups and several shapes of band tiles under the digests."""
    import _bigdigest as D
    if D.fixture_for(*D.BIG_SETS[name]) is None:
        pytest.skip("fixture tests/golden/mpcbig_%s.npz not generated" % name)
    z = D.load(name)
    seqs = D.seqs_of(name)
    import hashlib
    assert hashlib.sha256("\n".join(seqs).encode()).hexdigest() == str(z["seqs_sha"])
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs(seqs)
    g.calc_posteriors()
    assert D.compare_ea(z, g.get_ea()) is None
    g.build_store()
    assert D.compare_stage(z, 0, g) is None
    for it in range(2):
        g.cons_iter()
        g.cons_commit()
        assert D.compare_stage(z, it + 1, g) is None
    g.close()


def test_rdrp1000_sampled_reference_pin():
    """The real-data BENCH size (first 1000 rdrp records: 499 500 pairs, ~1.4e9 stored entries, relax through the band tiles the bench run
    cuts) against the compiled reference where a full reference run is days: stage A of ALL pairs (EA bits, stage-0 digests per 1000
    pairs) and MPCFlat::ConsPair (conspairflat.cpp:10-110) of iteration 1 for 2048 seeded pairs
    (tests/golden/mpcbig_rdrp1000_sampled.npz, tests/golden/make_golden.py big-sampled). bench.py repeats this check in its real_data leg."""
    import hashlib
    import _bigdigest as D
    from muscle_amd.synth import read_fasta
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rdrp_first1000.fa.gz")
    name = D.sampled_fixture_for_fasta(path, 1000)
    if name is None:
        pytest.skip("fixture tests/golden/mpcbig_rdrp1000_sampled.npz not generated")
    z = D.load(name)
    seqs = read_fasta(path)[:1000]
    assert hashlib.sha256("\n".join(seqs).encode()).hexdigest() == str(z["seqs_sha"])
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs(seqs)
    g.calc_posteriors()
    assert D.compare_ea(z, g.get_ea()) is None
    g.build_store()
    assert D.compare_stage(z, 0, g) is None
    g.cons_iter()
    g.cons_commit()
    assert D.compare_sample(z, g) is None
    info = g.relax_info()[0]
    assert "relax_band_kernel" in info, info
    # round 6: the SECOND iteration at this size too — the stage-2 matrices of the 15 pairs among 6 seeded sequences, which the
    # compiled reference reaches by relaxing only what those pairs read (tests/golden/make_golden.py big-stage2; the shortcut is
    # checked against two full ConsIter in tests/test_oracle_vs_ref.py)
    name2 = D.stage2_fixture_for_fasta(path, 1000)
    if name2 is not None:
        z2 = D.load(name2)
        assert str(z2["seqs_sha"]) == str(z["seqs_sha"])
        g.cons_iter()
        g.cons_commit()
        assert D.compare_stage2_clique(z2, g) is None
    g.close()
    assert name2 is not None, "fixture tests/golden/mpcbig_rdrp1000_stage2_clique.npz not generated (the stage-0 / stage-1 pins above passed)"


@pytest.mark.parametrize("nctx", [2, 3])
def test_group_of_contexts_equals_reference(nctx):
    """mpcgpu_group_* (several GPUs inside one process; here nctx contexts on device 0, peer-copy transport): after the
    sharded stage A + all-gather and after each sharded relax iteration + value all-gather, EVERY rank's store equals the
    reference's digests."""
    from muscle_amd._lib import MpcGroup
    g = G.mpc("n48_L260")
    grp = MpcGroup([0] * nctx)
    assert grp.transport() == "peer"
    grp.set_hmm(*G.hmm_tables())
    grp.set_seqs(g["seqs"])
    grp.calc_posteriors()
    views = [grp.ctx(r) for r in range(nctx)]
    for v in views:
        assert np.array_equal(P.bits(v.get_ea()), P.bits(g["ea"]))
        assert G.stage_digest(v.get_sparse_range()) == g["digest"][0]
    for it in range(2):
        grp.cons_iter()
        for v in views:
            assert G.stage_digest(v.get_sparse_range()) == g["digest"][it + 1]
    del views
    grp.close()


def test_group_of_eight_contexts_block_partition_config2_digests():
    """The partition of the 8-GPU run on ONE device (8 contexts, peer-copy transport): blocks of the pair triangle, partial stores
    (every context holds the matrices of half of the sequences), one-wave-per-pair commits with lazily refreshed packed values —
    BASELINE config 2 (256 x L~300), every stage of ranks 0, 3 and 7 against the compiled reference's digests; then the same with
    stage A in two pieces."""
    import os
    import _bigdigest as D
    from muscle_amd._lib import MpcGroup
    name = "n256_L300"
    z = D.load(name)
    seqs = D.seqs_of(name)
    for pieces in ("1", "2"):
        os.environ["MPCGPU_GROUP_PIECES"] = pieces
        try:
            grp = MpcGroup([0] * 8)
            grp.set_hmm(*G.hmm_tables())
            grp.set_seqs(seqs)
            grp.calc_posteriors()
            for r in (0, 3, 7):
                assert D.compare_ea(z, grp.ctx(r).get_ea()) is None, (pieces, r)
                assert D.compare_stage(z, 0, grp.ctx(r)) is None, (pieces, r)
            for it in range(2):
                grp.cons_iter()
                for r in (0, 3, 7):
                    assert D.compare_stage(z, it + 1, grp.ctx(r)) is None, (pieces, it, r)
            grp.close()
        finally:
            del os.environ["MPCGPU_GROUP_PIECES"]


def test_group_of_eight_contexts_config3_digests(monkeypatch):
    """BASELINE config 4's partition at its own size on one device: 1000 x L~400 over 8 contexts (peer copies), blocks + partial
    stores (500 of 1000 sequences per context); EA and the stage-0 / stage-2 stores of ranks 0 and 5, all 499 500 pairs, against the
    compiled reference's digests (tests/golden/mpcbig_n1000_L400.npz). Each context's stage-A scratch is held to 4 GB: eight
    contexts share the one device's memory here."""
    import _bigdigest as D
    from muscle_amd._lib import MpcGroup
    monkeypatch.setenv("MPCGPU_SCRATCH_GB", "4")
    name = "n1000_L400"
    z = D.load(name)
    grp = MpcGroup([0] * 8)
    grp.set_hmm(*G.hmm_tables())
    grp.set_seqs(D.seqs_of(name))
    grp.calc_posteriors()
    for r in (0, 5):
        assert D.compare_ea(z, grp.ctx(r).get_ea()) is None, r
        assert D.compare_stage(z, 0, grp.ctx(r)) is None, r
    grp.cons_iter()
    grp.cons_iter()
    for r in (0, 5):
        assert D.compare_stage(z, 2, grp.ctx(r)) is None, r
    grp.close()


def test_full_alphabet_stage_beyond_64k_of_lds():
    """~127 distinct byte values: emission tables of 65 KB, the dynamic-LDS attribute per device and kernel (round-5 advisor
    finding: no test ran a stage there) == the oracle"""
    P.check_full_alphabet(None)


def test_align_alns_batch():
    """the joins of a guide-tree level in two launches (mpcgpu_align_alns_batch) == the same joins one call at a time"""
    P.check_align_alns_batch(None)


def test_group_rccl_loader_one_device():
    """librccl is dlopen()ed by the group layer on first use; a one-device communicator on request checks that the library is
    found and ncclCommInitAll / ncclCommDestroy work on this box (the N > 1 exchange itself needs N GPUs)."""
    import os
    from muscle_amd._lib import MpcGroup
    os.environ["MPCGPU_GROUP_TRANSPORT"] = "rccl"
    try:
        grp = MpcGroup([0])
        assert grp.transport() == "rccl"
        seqs = make_family(5, 60, seed=8)
        grp.set_hmm(*G.hmm_tables())
        grp.set_seqs(seqs)
        grp.calc_posteriors()
        grp.cons_iter()
        got = grp.ctx(0).get_sparse_range()
        grp.close()
    finally:
        del os.environ["MPCGPU_GROUP_TRANSPORT"]
    want, _ = P.run_oracle(seqs, iters=1)
    for (o1, v1), (o2, v2) in zip(got, want[1]):
        assert np.array_equal(o1, o2) and np.array_equal(v1, v2)


def test_post_candidate_lists_with_gaps():
    """Both finishing kernels (and the multi-pass EA path) on synthetic candidate lists against the dense CalcAlnScoreFlat /
    FromPost: rows whose first cell lies beyond the EA frontier, empty rows, rows of > 64 cells (round-2 advisor finding)."""
    P.check_post_scores(None, trials=120)


@pytest.mark.parametrize("mode", ["rows", "sort"])
@pytest.mark.parametrize("name", P.BP_SETS)
def test_buildpost_vs_reference_golden(name, mode, monkeypatch):
    """Device BuildPost (mpcgpu_build_post: the matrix itself), AlignAlns and the PProg join against the compiled reference's
    own matrices / paths / scores for the same joins (tests/golden/bp_*.npz: plain, weighted, transposed access of
    buildpostflat.cpp:78-100, explicit pair lists of buildposterior3flat.cpp:19-85); both forms of the device BuildPost."""
    monkeypatch.setenv("MPCGPU_BP", mode)
    P.check_buildpost_golden(name)


def _device_count():
    import torch
    return torch.cuda.device_count()


def test_two_gpus_rccl_group_and_torchrun_bench():
    """Self-arming: runs wherever the box has >= 2 GPUs (the build boxes have one: skipped there, so the first multi-GPU box
    exercises it). (1) the product's one-process group (mpcgpu_group_*, RCCL point-to-point sends / receives over xGMI) on
    devices 0 and 1: every rank's store equals the reference's digests for BASELINE config 2 after every stage; (2) the driver's
    launch of bench.py on 2 ranks (torch.distributed.run, backend nccl = RCCL): parity_digest must say match."""
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    import json
    import subprocess
    import sys
    import _bigdigest as D
    from muscle_amd._lib import MpcGroup
    name = "n256_L300"
    z = D.load(name)
    seqs = D.seqs_of(name)
    s, t, m, i, thr = G.hmm_tables()
    grp = MpcGroup([0, 1])
    assert grp.transport() == "rccl"
    grp.set_hmm(s, t, m, i, thr)
    grp.set_seqs(seqs)
    grp.calc_posteriors()
    for r in range(2):
        assert D.compare_ea(z, grp.ctx(r).get_ea()) is None, r
        assert D.compare_stage(z, 0, grp.ctx(r)) is None, r
    for it in range(2):
        grp.cons_iter()
        for r in range(2):
            assert D.compare_stage(z, it + 1, grp.ctx(r)) is None, (it, r)
    grp.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", "--nseqs", "256", "--seqlen", "300", "--steps", "1",
                          "--warmup", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads(lines[-1])
    assert rec["n_gpus"] == 2 and rec["parity_digest"] == "match", rec
    assert rec["exchange_ms"] > 0
    mg = rec["multi_gpu"]  # round 6: the block partition, and what the line says about the run itself
    assert mg["rccl_ranks"] == 2 and mg["backend"] == "nccl" and "blocks of the pair triangle" in mg["partition"], mg
    assert len(mg["phase_ms_per_rank"]) == 2 and sum(mg["pairs_per_rank"]) == 256 * 255 // 2 and len(set(mg["devices"])) >= 1, mg
    # eight ranks' partition (blocks, partial stores: half of the sequences per rank) on the two devices, four contexts each
    grp = MpcGroup([0, 1, 0, 1, 0, 1, 0, 1])
    grp.set_hmm(s, t, m, i, thr)
    grp.set_seqs(seqs)
    grp.calc_posteriors()
    for it in range(2):
        grp.cons_iter()
    for r in (0, 5):
        assert D.compare_ea(z, grp.ctx(r).get_ea()) is None and D.compare_stage(z, 2, grp.ctx(r)) is None, r
    grp.close()


def test_align_pairs_list_longer_than_a_chunk():
    """mpcgpu_get_list_sparse after a list that ran in two stage-A chunks: the caller's index, not the last chunk's"""
    P.check_align_pairs_chunked()


def test_align_pairs_vs_reference_golden():
    """mpcgpu_align_pairs / mpcgpu_get_list_sparse against the compiled reference's AlignPairFlat_SparsePost
    (alignpairflat.cpp:3-27; callers uclust.cpp:14, transaln.cpp:787, eadistmx.cpp:54): path, EA bits, FromPost matrix."""
    P.check_align_pairs_golden("ap_ragged")


@pytest.mark.gpu
def test_fb_chains():
    """pairs that share their row sequence swept back to back (kernels_fbc.h) == one pair per sweep == the oracle"""
    P.check_fb_chains()
