"""Kernel-logic check WITHOUT a GPU: the product sources (muscle_amd/csrc) compiled against the
SIMT emulator (tests/emu) must agree bit for bit with the oracle on tiny inputs. This validates
indexing, the systolic skew schedule, border cells, sorting/EA/sparsify, slab build, relax and
commit before GPU minutes are spent; the real parity tests are the -m gpu ones."""
import os
import subprocess

import numpy as np
import pytest

import _golden as G
import _parity as P
from muscle_amd._lib import MpcGpu
from muscle_amd.synth import make_family

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
EMU_LIB = os.path.join(EMU_DIR, "libmpcgpu_emu.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-C", EMU_DIR], stdout=subprocess.DEVNULL)
    return EMU_LIB


@pytest.mark.parametrize("name", ["n2_L40", "n3_L30"])
def test_emu_vs_golden_small(emu, name):
    g = G.mpc(name)
    stages, ea = P.run_lib(g["seqs"], lib_path=emu)
    assert np.array_equal(P.bits(ea), P.bits(g["ea"]))
    for s in range(g["nstages"]):
        assert G.stage_digest(stages[s]) == g["digest"][s]


def test_emu_vs_oracle_ragged(emu):
    seqs = ["M", "MKVLA", make_family(1, 70, seed=9)[0], "ACDEFGHIKLMNPQRSTVWY" * 4, "WWWWWWWW"]
    P.assert_same(P.run_lib(seqs, lib_path=emu), P.run_oracle(seqs), "ragged")


def test_emu_multirow_lanes(emu):
    # LX > 64 -> H = 2..3 rows per lane, several lanes idle at the tail
    seqs = make_family(3, 150, seed=4)
    P.assert_same(P.run_lib(seqs, lib_path=emu), P.run_oracle(seqs), "H>1")


def _with_env(env, fn):
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


def test_emu_relax_many_tiles(emu):
    # 9 sequences -> 3x3 blocks of 4: diagonal, off-diagonal and ragged edge tiles of the LDS-tiled relax
    seqs = make_family(9, 18, seed=5)
    want = P.run_oracle(seqs)
    P.assert_same(P.run_lib(seqs, lib_path=emu), want, "tiled 4x4")
    P.assert_same(_with_env({"MPCGPU_RELAX": "gather"}, lambda: P.run_lib(seqs, lib_path=emu)), want, "gather")


@pytest.mark.parametrize("lds_kb", [2, 1])
def test_emu_relax_small_lds_shapes(emu, lds_kb):
    # a tiny LDS budget forces the 4x2 / 2x2 / 2x1 / 1x1 tile shapes (or the gather fallback)
    seqs = make_family(6, 18, seed=6)
    got = _with_env({"MPCGPU_RELAX_LDS_KB": str(lds_kb)}, lambda: P.run_lib(seqs, lib_path=emu))
    P.assert_same(got, P.run_oracle(seqs), "lds %d KB" % lds_kb)


def test_emu_relax_tile_splitting(emu):
    # a slot budget of 3 makes the host split every 4x4 block recursively (Y range, then X range)
    seqs = make_family(8, 16, seed=8)
    got = _with_env({"MPCGPU_RELAX_SLOTS": "3"}, lambda: P.run_lib(seqs, lib_path=emu))
    P.assert_same(got, P.run_oracle(seqs), "split tiles")


def test_emu_relax_two_slots_per_pair(emu):
    # nnz > 1024 -> NENT = 2: one pair spans two register slots, matrices are staged in two passes
    # (low-complexity repeats spread the posterior over many diagonals: ~3.5 stored cells per row)
    seqs = ["AC" * 165, "AC" * 162 + "A", "CA" * 164]
    stages, ea = P.run_lib(seqs, lib_path=emu)
    assert max(len(v) // 2 for _, v in stages[0]) > 1024
    P.assert_same((stages, ea), P.run_oracle(seqs), "nent=2")


@pytest.mark.parametrize("kernel", ["by size", "one wave", "waves", "lds rows"])
def test_emu_calc_aln(emu, kernel, monkeypatch):
    """CalcAlnFlat + TraceBackFlat on the device (kernels_aln.h) vs the oracle: same path string,
    same score bits, including tie cases (equal B/X/Y candidates) and general non-posterior input."""
    # calc_aln_wave_kernel (<= 512 columns) / calc_aln_quad_kernel (<= 4096 columns) / calc_aln_kernel; a kernel that cannot take
    # a matrix leaves it to the next one
    monkeypatch.setenv("MPCGPU_ALN_KERNEL", str(["by size", "one wave", "waves", "lds rows"].index(kernel)))
    import _oracle as O
    from muscle_amd._lib import MpcGpu
    rng = np.random.default_rng(5)
    g = MpcGpu(0, emu)
    mats = []
    for LX, LY in ((1, 1), (1, 7), (9, 1), (13, 17), (40, 33)):
        P0 = (rng.random((LX, LY)) < 0.15) * rng.random((LX, LY))
        mats.append(P0.astype(np.float32))
        mats.append(np.round(P0 * 4).astype(np.float32) / 4)  # many exact ties
    # the one-wave kernel picks its columns per lane by width (4 .. 8: kernels_aln.h): both sides of every class boundary, with ties
    for LY in (255, 256, 319, 320, 383, 384, 447, 448, 511):
        P0 = (rng.random((11, LY)) < 0.1) * rng.random((11, LY))
        mats.append((np.round(P0 * 4) / 4).astype(np.float32))
    mats.append(np.zeros((5, 6), np.float32))                 # all ties: every cell equal
    mats.append((rng.random((30, 1100)) * 3).astype(np.float32))  # more columns than threads
    if kernel == "waves":  # more rows than one staged traceback block holds (150 KB / 320 bytes per row)
        mats.append(((rng.random((520, 1100)) < 0.01) * np.round(rng.random((520, 1100)) * 4) / 4).astype(np.float32))
    for M in mats:
        path, sc = g.calc_aln(M)
        sc0, path0 = O.calc_aln(M)
        assert path == path0, M.shape
        assert P.bits(sc) == P.bits(sc0)
    g.close()


def test_emu_post_kernels(emu):
    """Both post kernels (row lists: default; bitonic sort: general path) and the multi-pass EA path of
    the row-list kernel (forced with a batch of 3 cells) give the oracle's store and EA."""
    seqs = ["AC" * 30, "CA" * 28, "ACAC" * 13 + "A", make_family(1, 50, seed=3)[0], "M"]
    want = P.run_oracle(seqs)
    P.assert_same(P.run_lib(seqs, lib_path=emu), want, "rows")
    P.assert_same(_with_env({"MPCGPU_POST": "sort"}, lambda: P.run_lib(seqs, lib_path=emu)), want, "sort")
    P.assert_same(_with_env({"MPCGPU_POST_BATCH": "3"}, lambda: P.run_lib(seqs, lib_path=emu)), want, "batch 3")
    P.assert_same(_with_env({"MPCGPU_POST_SORT_CAP": "8"}, lambda: P.run_lib(seqs, lib_path=emu)), want, "global scratch list")


def test_emu_align_alns(emu):
    """Device BuildPost + CalcAlnFlat (mpcgpu_align_alns) vs the numpy restatement of
    buildpostflat.cpp + the oracle's CalcAlnFlat: same path, same score bits; both orientations
    (s<t and s>t), after two relax iterations (current probabilities of the store)."""
    import _buildpost as BP
    import _oracle as O
    from muscle_amd._lib import MpcGpu
    rng = np.random.default_rng(3)
    seqs = make_family(7, 22, seed=23)
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0, emu)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs(seqs)
    g.calc_posteriors()
    g.build_store()
    for _ in range(2):
        g.cons_iter()
        g.cons_commit()
    stage = g.get_sparse_range()
    pidx = {p: k for k, p in enumerate((a, b) for a in range(7) for b in range(a + 1, 7))}
    for grp1, grp2 in (([0, 2, 5], [1, 3, 4, 6]), ([6, 1], [0]), ([3], [2]), ([4, 5, 6], [0, 1, 2, 3])):
        rows1, C1 = BP.random_msa(seqs, grp1, rng)
        rows2, C2 = BP.random_msa(seqs, grp2, rng)
        m1 = [BP.pos_to_col(r) for r in rows1]
        m2 = [BP.pos_to_col(r) for r in rows2]
        post = BP.build_post(stage, pidx, grp1, grp2, m1, m2, C1, C2)
        sc0, path0 = O.calc_aln(post)
        path, sc = g.align_alns(grp1, grp2, m1, m2, C1, C2)
        assert path == path0 and P.bits(sc) == P.bits(sc0), (grp1, grp2)
        # sequence weights (buildpostflat.cpp:41,52,74: every contribution is (w1*w2)*P)
        w1 = rng.uniform(0.2, 1.8, len(grp1)).astype(np.float32)
        w2 = rng.uniform(0.2, 1.8, len(grp2)).astype(np.float32)
        sc0w, path0w = O.calc_aln(BP.build_post(stage, pidx, grp1, grp2, m1, m2, C1, C2, w1, w2))
        pathw, scw = g.align_alns(grp1, grp2, m1, m2, C1, C2, w1, w2)
        assert pathw == path0w and P.bits(scw) == P.bits(sc0w), ("weighted", grp1, grp2)
    # timing off (mpcgpu_timers_enable: what the drop-in runs with): same results, nothing measured
    g.timers_reset()
    g.timers_enable(False)
    path, sc = g.align_alns(grp1, grp2, m1, m2, C1, C2)
    assert path == path0 and P.bits(sc) == P.bits(sc0)
    assert all(v == (0.0, 0) for v in g.timers_get().values())
    g.timers_enable(True)
    g.align_alns(grp1, grp2, m1, m2, C1, C2)
    assert g.timers_get()["calc_aln"][1] == 1
    g.close()


def test_emu_align_alns_batch(emu):
    """the joins of a guide-tree level in two launches (mpcgpu_align_alns_batch) == the same joins one call at a time"""
    P.check_align_alns_batch(emu)


def test_emu_align_alns_long_runs(emu):
    """Runs of the in-order reduction longer than one chunk (64 terms) and than one group of chunks (512): 24 x 24 closely
    related sequences, so the cells on the path collect a term from most of the 1024 pairs."""
    import _buildpost as BP
    import _oracle as O
    from muscle_amd._lib import MpcGpu
    rng = np.random.default_rng(9)
    n = 64
    seqs = make_family(n, 14, seed=31, p_del=0.01, p_ins=0.01, p_sub=0.05)
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0, emu)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs(seqs)
    g.calc_posteriors()
    g.build_store()
    stage = g.get_sparse_range()
    pidx = {p: k for k, p in enumerate((a, b) for a in range(n) for b in range(a + 1, n))}
    grp1, grp2 = list(range(0, n, 2)), list(range(1, n, 2))  # both orientations of the stored pairs occur
    rows1, C1 = BP.random_msa(seqs, grp1, rng)
    rows2, C2 = BP.random_msa(seqs, grp2, rng)
    m1 = [BP.pos_to_col(r) for r in rows1]
    m2 = [BP.pos_to_col(r) for r in rows2]
    post = BP.build_post(stage, pidx, grp1, grp2, m1, m2, C1, C2)
    # the longest run: contributions to one cell (each stored entry of each cross pair lands in exactly one)
    cnt = np.zeros((C1, C2), np.int64)
    for a, S in enumerate(grp1):
        for b, T in enumerate(grp2):
            off, val = stage[pidx[(min(S, T), max(S, T))]]
            col = val[1::2]
            for i in range(len(off) - 1):
                for k in range(off[i], off[i + 1]):
                    if S < T:
                        cnt[m1[a][i], m2[b][col[k]]] += 1
                    else:
                        cnt[m1[a][col[k]], m2[b][i]] += 1
    assert cnt.max() > 512 and (cnt > 64).sum() > 10
    sc0, path0 = O.calc_aln(post)
    path, sc = g.align_alns(grp1, grp2, m1, m2, C1, C2)
    assert path == path0 and P.bits(sc) == P.bits(sc0)
    g.close()


def test_emu_align_msas(emu):
    """PProg's MSA x MSA join on the device (mpcgpu_align_msas: stage A on an explicit pair list, both
    index orders, + CalcPosteriorFlat3 + CalcAlnFlat) vs the oracle per pair and a numpy restatement
    of buildposterior3flat.cpp:19-85 (contributions in pair-list order)."""
    import _buildpost as BP
    import _oracle as O
    from muscle_amd._lib import MpcGpu
    rng = np.random.default_rng(9)
    seqs = make_family(8, 20, seed=31)
    s, t, m, i, thr = G.hmm_tables()
    h = O.make_hmm(s, t, m, i)
    g = MpcGpu(0, emu)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs_registry(seqs)
    grp1, grp2 = [5, 0, 3], [7, 1, 2, 6, 4]
    rows1, C1 = BP.random_msa(seqs, grp1, rng)
    rows2, C2 = BP.random_msa(seqs, grp2, rng)
    pairs = [(a, b) for a in range(len(grp1)) for b in range(len(grp2)) if (a + b) % 4 != 3]  # a sampled subset
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


# ---- structure-profile ("mega") emissions: fb_kernel<H, true> + mega_prepare_kernel -------------------------
@pytest.mark.parametrize("name", ["mega_synth_3x25_s5_f3", "mega_synth_2x70_s7"])
def test_emu_mega_vs_golden(emu, name):
    """the reference's own parsed tables and outputs (calcpost.cpp:14-22 branch)"""
    m = G.mega(name)
    stages, ea = P.run_lib(m["seqs"], lib_path=emu, mega=m)
    assert np.array_equal(P.bits(ea), P.bits(m["ea"]))
    for s in range(m["nstages"]):
        assert G.stage_digest(stages[s]) == m["digest"][s]


def test_emu_mega_vs_oracle(emu):
    # ragged lengths incl. length 1 and H = 2 rows per lane; random tables with all 8 features
    seqs = ["M", "MKVLA", make_family(1, 80, seed=3)[0], make_family(1, 33, seed=4)[0]]
    mega = P.random_mega(seqs, seed=11)
    P.assert_same(P.run_lib(seqs, lib_path=emu, mega=mega), P.run_oracle(seqs, mega=mega), "mega ragged")
    # and the switch back to byte sequences on the same context state is covered by every other test


def test_emu_mega_rejects_bad_input(emu):
    from muscle_amd._lib import MpcGpu
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0, emu)
    g.set_hmm(s, t, m, i, thr)
    seqs = ["MKVLA", "MKILA"]
    g.set_seqs(seqs)
    mega = P.random_mega(seqs, seed=2, nfeat=3)
    mega["profs"][1][4] = 200  # letter outside its alphabet
    with pytest.raises(RuntimeError, match="outside its alphabet"):
        g.set_mega(mega["alpha"], mega["weight"], mega["lp"], mega["mx"], mega["profs"])
    g.close()


def test_emu_set_seqs_takes_every_seven_bit_byte(emu):
    """the alphabet: all 128 seven-bit values are accepted (the compacted tables grow to (A*A + A) floats of LDS; the golden set
    alpha82 pins the arithmetic of such an input against the compiled reference); a byte >= 128 is refused, by position"""
    from muscle_amd._lib import MpcGpu
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0, emu)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs([bytes(range(1, 65)), bytes(range(65, 128)) + b"A"])  # 127 distinct values
    with pytest.raises(RuntimeError, match=r"non-ASCII byte 200 at position 3"):
        g.set_seqs([b"ACD" + bytes([200]), b"ACDE"])
    g.close()


def test_emu_full_alphabet_stage(emu):
    """the whole stage over ~127 distinct byte values (65 KB of emission tables in LDS) == the oracle"""
    P.check_full_alphabet(emu)


def test_emu_mega_then_letters_on_one_context(emu):
    """set_seqs drops the profiles: the same context goes back to letter emissions"""
    from muscle_amd._lib import MpcGpu
    s, t, m, i, thr = G.hmm_tables()
    seqs = make_family(3, 24, seed=31)
    mega = P.random_mega(seqs, seed=5, nfeat=4)
    g = MpcGpu(0, emu)
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
    assert not np.array_equal(P.bits(ea_mega), P.bits(ea_plain))


# ---- row blocks: fb_kernel<H, MEGA, LONG=true> (sequences X longer than 64*MPC_HMAX rows) -------------------
# H = 1 and a low threshold reach several 64-row blocks with short sequences (the GPU tests use real lengths).
def test_emu_row_blocks(emu):
    seqs = [make_family(1, 131, seed=21)[0], make_family(1, 66, seed=22)[0], make_family(1, 64, seed=24)[0], "MKV"]
    # every pair through the row-block kernel: 3, 2 and 1 blocks (LX = 131, 66, 64, 3)
    got = _with_env({"MPCGPU_FB_LONG_H": "1", "MPCGPU_FB_LONG_MIN": "2"}, lambda: P.run_lib(seqs, lib_path=emu))
    P.assert_same(got, P.run_oracle(seqs), "row blocks")


def test_emu_row_blocks_general_post_and_mega(emu):
    seqs = [make_family(1, 130, seed=31)[0], make_family(1, 66, seed=33)[0]]
    env = {"MPCGPU_FB_LONG_H": "1", "MPCGPU_FB_LONG_MIN": "65"}
    got = _with_env(dict(env, MPCGPU_POST="sort"), lambda: P.run_lib(seqs, lib_path=emu))
    P.assert_same(got, P.run_oracle(seqs), "row blocks + general post kernel")
    mega = P.random_mega(seqs, seed=9)
    got = _with_env(env, lambda: P.run_lib(seqs, lib_path=emu, mega=mega))
    P.assert_same(got, P.run_oracle(seqs, mega=mega), "row blocks + mega")


def test_emu_row_blocks_real_lengths(emu):
    """the shipped configuration: LX >= 1025 -> fb_kernel<7, MEGA, LONG> with 448-row blocks (3 and 4 blocks here)"""
    seqs = [make_family(1, 1400, seed=51)[0], make_family(1, 1030, seed=52)[0], make_family(1, 200, seed=53)[0]]
    P.assert_same(P.run_lib(seqs, lib_path=emu), P.run_oracle(seqs), "row blocks, real lengths")


# ---- host-side paths that only special sizes reach -------------------------------------------------------------
def test_emu_candidate_overflow_retry(emu):
    """a pair with more candidates than the buffer (floor 1024 entries) makes the batch run again with twice the room"""
    seqs = make_family(3, 700, seed=61)
    got = _with_env({"MPCGPU_CAND_PER_ROW": "1"}, lambda: P.run_lib(seqs, lib_path=emu))
    P.assert_same(got, P.run_oracle(seqs), "overflow retry")


def test_emu_one_pair_per_batch(emu):
    """no scratch budget -> every pair is its own stage-A batch: the packed shard grows batch by batch"""
    seqs = make_family(5, 40, seed=62) + ["MKV"]
    got = _with_env({"MPCGPU_SCRATCH_GB": "0"}, lambda: P.run_lib(seqs, lib_path=emu))
    P.assert_same(got, P.run_oracle(seqs), "one pair per batch")


def test_emu_relax_512_thread_workgroups(emu):
    seqs = make_family(7, 24, seed=63)
    got = _with_env({"MPCGPU_RELAX_WG": "512"}, lambda: P.run_lib(seqs, lib_path=emu))
    P.assert_same(got, P.run_oracle(seqs), "512-thread relax workgroups")


@pytest.mark.parametrize("env", [{"MPCGPU_RELAX_WG": "1024"}, {"MPCGPU_RELAX_WG": "1024", "MPCGPU_RELAX_NBUF": "1"}, {"MPCGPU_RELAX_WG": "2048"},
                                 {"MPCGPU_RELAX_WG": "512", "MPCGPU_RELAX_NBUF": "2"}, {"MPCGPU_RELAX_SLOTS": "3"}, {"MPCGPU_RELAX_LDS_KB": "6"},
                                 {"MPCGPU_RELAX_WG": "1024", "MPCGPU_RELAX_LDS_KB": "12"}])
def test_emu_relax_var_geometries(emu, env):
    """relax_var_kernel in its other shapes: one 1024-thread workgroup per CU with two staging buffers (DMA of step Z+1 under the
    merges of step Z) or one, two 1024- or 512-thread workgroups per CU, a small slot budget and a small LDS (tiles split by
    the host's per-tile fit check) — all the default's arithmetic, all bit-identical to the oracle."""
    seqs = make_family(9, 18, seed=5) + [make_family(1, 70, seed=9)[0], "MKV"]
    got = _with_env(env, lambda: P.run_lib(seqs, lib_path=emu))
    P.assert_same(got, P.run_oracle(seqs), "relax_var_kernel %s" % env)


@pytest.mark.parametrize("env", [{}, {"MPCGPU_RELAX_WG": "1024"}, {"MPCGPU_RELAX_WG": "1024", "MPCGPU_RELAX_NBUF": "1"}, {"MPCGPU_RELAX_WG": "768"}])
def test_emu_relax_staging_with_late_dma(emu, env):
    """EMU_DMA=late: an LDS-DMA transfer poisons its 16 destination bytes when it is issued and delivers them only when the issuing
    thread waits for it (mpc_dma_wait) — as late as the hardware may. A walk that merged a record before the wait + barrier that
    follows its staging (one staging buffer), or that let step Z+1's transfers into a buffer step Z still reads (two buffers),
    computes with garbage: the results must still be the oracle's."""
    seqs = make_family(9, 18, seed=5) + [make_family(1, 70, seed=9)[0], "MKV"]
    got = _with_env(dict(env, EMU_DMA="late"), lambda: P.run_lib(seqs, lib_path=emu))
    want = P.run_oracle(seqs)
    P.assert_same(got, want, "late DMA %s" % env)
    if not env:  # the checker checks: when the waits deliver nothing the same run must NOT reproduce the oracle
        bad = _with_env({"EMU_DMA": "never"}, lambda: P.run_lib(seqs, lib_path=emu))
        with pytest.raises(AssertionError):
            P.assert_same(bad, want, "transfers never delivered")


def test_emu_dense_records_long_rows(emu):
    """rows with many entries (weakly related sequences: several blocks per row chained through the overflow region)"""
    seqs = make_family(6, 60, seed=8, p_sub=0.7) + make_family(2, 50, seed=9, p_sub=0.6)
    P.assert_same(P.run_lib(seqs, lib_path=emu), P.run_oracle(seqs), "dense records, long rows")


def test_emu_row_blocks_vs_reference_golden(emu):
    """the reference's own stage outputs for three sequences of ~1100 residues (X longer than 1024 rows)"""
    g = G.mpc("n3_L1100")
    assert min(len(s) for s in g["seqs"]) > 1024
    stages, ea = P.run_lib(g["seqs"], lib_path=emu)
    assert np.array_equal(P.bits(ea), P.bits(g["ea"]))
    for s in range(g["nstages"]):
        assert G.stage_digest(stages[s]) == g["digest"][s]


def test_emu_very_long_row_sequence(emu):
    """a 9048-residue row sequence against short ones: 21 row blocks in the fb kernel, 16-bit-column candidate keys,
    and (longer than 4095) the gather relax instead of the LDS tiles"""
    fam = make_family(2, 50, seed=3)
    big = make_family(1, 9000, seed=4)[0]
    big = big[:4000] + fam[0] + big[4000:]  # related to the short ones somewhere in the middle
    seqs = [big, fam[0], fam[1]]
    P.assert_same(P.run_lib(seqs, lib_path=emu), P.run_oracle(seqs), "9048-long row sequence")


@pytest.mark.parametrize("order", ["reverse", "random"])
def test_emu_other_thread_orders(emu, order):
    """race check: the emulator runs the GPU threads of a block in reverse / shuffled order (tests/_emu_sched_check.py)"""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, EMU_SCHED=order, PYTHONPATH=os.path.dirname(here) + os.pathsep + here)
    r = subprocess.run([sys.executable, "-u", os.path.join(here, "_emu_sched_check.py")], env=env, cwd=here,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert r.returncode == 0 and "OK thread order " + order in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("nctx", [2, 3])
def test_emu_group_of_contexts(nctx):
    """mpcgpu_group_* on the emulator: pairs sharded over nctx contexts, all-gather of the packed shards, sharded relax,
    all-gather of the values — every rank's store equals the oracle's after every stage."""
    from muscle_amd._lib import MpcGroup
    seqs = make_family(9, 110, seed=3) + make_family(2, 200, seed=4)
    want = P.run_oracle(seqs)
    grp = MpcGroup([0] * nctx, EMU_LIB)
    grp.set_hmm(*G.hmm_tables())
    grp.set_seqs(seqs)
    grp.calc_posteriors()
    views = [grp.ctx(r) for r in range(nctx)]
    st = [[v.get_sparse_range()] for v in views]
    ea = [v.get_ea().copy() for v in views]
    for _ in range(2):
        grp.cons_iter()
        for r, v in enumerate(views):
            st[r].append(v.get_sparse_range())
    for r in range(nctx):
        P.assert_same((st[r], ea[r]), want, "group of %d, rank %d" % (nctx, r))
    del views
    grp.close()


@pytest.mark.parametrize("nctx,n,pieces", [(3, 12, 2), (4, 16, 3), (8, 16, 2)])
def test_emu_group_block_partition_partial_stores(nctx, n, pieces, monkeypatch):
    """The block partition with real kernels: every context owns blocks of the pair triangle, enumerates the pairs
    block by block, runs stage A in pieces, imports the shards piece by piece and builds a PARTIAL store (the records of its
    blocks' sequences only). After every stage every rank's packed matrices — read back in InitPairs order — equal the oracle's;
    afterwards BuildPost on rank 0 (which completes its store first) equals BuildPost on a single context."""
    from muscle_amd._lib import MpcGroup, MpcGpu, plan_partition
    monkeypatch.setenv("MPCGPU_GROUP_PIECES", str(pieces))
    seqs = make_family(n - 3, 44, seed=5) + make_family(3, 70, seed=6)
    rects, pos = plan_partition([len(s) for s in seqs], nctx, EMU_LIB)
    assert len(rects) > 0  # blocks, not the contiguous fallback
    want = P.run_oracle(seqs)
    grp = MpcGroup([0] * nctx, EMU_LIB)
    grp.set_hmm(*G.hmm_tables())
    grp.set_seqs(seqs)
    grp.calc_posteriors()
    views = [grp.ctx(r) for r in range(nctx)]
    st = [[v.get_sparse_range()] for v in views]
    ea = [v.get_ea().copy() for v in views]
    for _ in range(2):
        grp.cons_iter()
        for r, v in enumerate(views):
            st[r].append(v.get_sparse_range())
    for r in range(nctx):
        P.assert_same((st[r], ea[r]), want, "block partition over %d, rank %d" % (nctx, r))
    # a join on rank 0's (partial, then completed) store against the same join on one context
    one = MpcGpu(0, EMU_LIB)
    one.set_hmm(*G.hmm_tables())
    one.set_seqs(seqs)
    one.calc_posteriors()
    one.build_store()
    for _ in range(2):
        one.cons_iter()
        one.cons_commit()
    s1, s2 = [0, n - 1, 3], [5, 1, n - 2, 7]
    p2c1 = [np.arange(len(seqs[i]), dtype=np.uint32) for i in s1]
    p2c2 = [np.arange(len(seqs[i]), dtype=np.uint32) for i in s2]
    C1, C2 = max(len(seqs[i]) for i in s1), max(len(seqs[i]) for i in s2)
    a = views[0].build_post(s1, s2, p2c1, p2c2, C1, C2)
    b = one.build_post(s1, s2, p2c1, p2c2, C1, C2)
    assert np.array_equal(P.bits(a.ravel()), P.bits(b.ravel()))
    monkeypatch.setenv("MPCGPU_BP", "sort")  # the general form looks the pairs' entry counts up by position in the pair order
    a2 = views[1].build_post(s1, s2, p2c1, p2c2, C1, C2)
    assert np.array_equal(P.bits(a2.ravel()), P.bits(b.ravel()))
    one.close()
    del views
    grp.close()


def test_emu_sharded_context_lazy_packed_values(emu):
    """A context that relaxes a part of the pairs writes the packed matrices of ITS pairs at a commit and refreshes the others' from the
    values array when somebody reads them — but only the ranges of entries that HAVE been committed: after a relax + a commit of the own
    slice alone, a foreign pair's matrix must still be the stage-0 one, the own pairs' the relaxed ones."""
    from muscle_amd._lib import MpcGpu
    seqs = make_family(9, 30, seed=41)
    want = P.run_oracle(seqs, iters=1)[0]
    g = MpcGpu(0, emu)
    g.set_hmm(*G.hmm_tables())
    g.set_seqs(seqs)
    npairs = g.npairs
    g.calc_posteriors(0, npairs)
    nb, ptr = g.shard_info()
    buf = np.zeros(nb + 16, np.uint8)
    g.shard_export(buf.ctypes.data)
    k0, k1 = 5, 17  # this context's own range
    g.store_import_part([0], [npairs], [nb], [0], buf.ctypes.data, k0, k1)
    g.cons_iter(k0, k1)
    first, count = g.values_slice(k0, k1)
    g.cons_commit_range(first, count)
    got = g.get_sparse_range()
    for k in range(npairs):
        o, v = got[k]
        wo, wv = want[1][k] if k0 <= k < k1 else want[0][k]
        assert np.array_equal(o, wo) and np.array_equal(v, wv), k
    g.close()


@pytest.mark.parametrize("n,nctx", [(3, 4), (2, 2), (4, 3)])
def test_emu_group_more_contexts_than_work(n, nctx):
    """mpcgpu_group_* when shards are empty or tiny (more contexts than pairs; two sequences: no consistency stage)"""
    from muscle_amd._lib import MpcGroup
    seqs = make_family(n, 40, seed=3)
    want = P.run_oracle(seqs)
    grp = MpcGroup([0] * nctx, EMU_LIB)
    grp.set_hmm(*G.hmm_tables())
    grp.set_seqs(seqs)
    grp.calc_posteriors()
    views = [grp.ctx(r) for r in range(nctx)]
    st = [[v.get_sparse_range()] for v in views]
    ea = [v.get_ea().copy() for v in views]
    if n >= 3:  # mpcflat.cpp:176
        for _ in range(2):
            grp.cons_iter()
            for r, v in enumerate(views):
                st[r].append(v.get_sparse_range())
    for r in range(nctx):
        P.assert_same((st[r], ea[r]), want, "n=%d, %d contexts, rank %d" % (n, nctx, r))
    del views
    grp.close()


def test_emu_post_candidate_lists_with_gaps(emu):
    """EA frontier of post_rows_kernel: columns between the frontier and a row's first cell must receive the flat suffix
    (round-2 advisor finding); both finishing kernels against the dense DP on synthetic candidate lists."""
    P.check_post_scores(emu, trials=36)


@pytest.mark.parametrize("mode", ["rows", "sort"])
@pytest.mark.parametrize("name", P.BP_SETS)
def test_emu_buildpost_vs_reference_golden(emu, name, mode):
    """mpcgpu_build_post / align_alns / align_msas against matrices, paths and scores generated by the compiled reference; both
    forms of the device BuildPost (one launch by rows for joins of few pairs / records, stable sort, in-order run sums)."""
    _with_env({"MPCGPU_BP": mode}, lambda: P.check_buildpost_golden(name, emu))


def test_emu_align_pairs_vs_reference_golden(emu):
    """AlignPairFlat on the device against the compiled reference's AlignPairFlat_SparsePost (alignpairflat.cpp:3-27)."""
    P.check_align_pairs_golden("ap_ragged", emu)


def test_emu_align_pairs_list_longer_than_a_chunk(emu):
    """mpcgpu_get_list_sparse indexes the CALLER'S list: a list of 300 pairs runs in two stage-A chunks (256 + 44), and the matrix of a pair
    of the first chunk must still be that pair's (the index used to address the last chunk — another pair's record, or an error)"""
    P.check_align_pairs_chunked(emu)


def test_emu_fb_chains(emu):
    """pairs that share their row sequence swept back to back (kernels_fbc.h) == one pair per sweep == the oracle"""
    P.check_fb_chains(emu)


def test_emu_relax_two_geometries(emu):
    """pairs whose records do not fit the two-workgroups-per-CU geometry go to a second launch of the one-workgroup geometry
    (real data: rdrp records of 24..43 KB): a 1 KB budget for the first and 8 KB for the second, sequences of 12..60 residues"""
    seqs = make_family(4, 12, seed=31) + make_family(3, 60, seed=32) + make_family(3, 30, seed=33)
    want = P.run_oracle(seqs)
    info = {}

    def run():
        s, t, m, i, thr = G.hmm_tables()
        g = MpcGpu(0, emu)
        g.set_hmm(s, t, m, i, thr)
        g.set_seqs(seqs)
        g.calc_posteriors()
        g.build_store()
        g.cons_iter()
        info["geo"] = g.relax_info()[0]
        g.close()
        return P.run_lib(seqs, lib_path=emu)
    got = _with_env({"MPCGPU_RELAX_TILES": "pairs", "MPCGPU_RELAX_LDS_KB": "1", "MPCGPU_RELAX_LDS_KB_1024": "8"}, run)
    assert "second launch" in info["geo"] and "+ 1 x 1024" in info["geo"], info["geo"]
    P.assert_same(got, want, "two geometries")
    got0 = _with_env({"MPCGPU_RELAX_TILES": "pairs", "MPCGPU_RELAX_LDS_KB": "1", "MPCGPU_RELAX_LDS_KB_1024": "8", "MPCGPU_RELAX_MIXED": "0"}, lambda: P.run_lib(seqs, lib_path=emu))
    P.assert_same(got0, want, "one-workgroup geometry alone")


# ---- band tiles (relax_band_kernel, kernels_relaxb.h) ----------------------------------------------------------------------
@pytest.mark.parametrize("env", [
    {"MPCGPU_RELAX_SHAPE": "8,8"},                                   # several row bands per super-tile come from the slot budget
    {"MPCGPU_RELAX_SHAPE": "8,8", "MPCGPU_RELAX_SLOTS": "1"},        # many bands, tiles split further (band, Y, X)
    {"MPCGPU_RELAX_SHAPE": "4,2", "MPCGPU_RELAX_SLOTS": "2"},
    {"MPCGPU_RELAX_SHAPE": "1,1"},
    {"MPCGPU_RELAX_LDS_KB": "10"},                                   # the shape search ends in "one step resident": every step staged after the merges
    {"MPCGPU_RELAX_LDS_KB": "8"},                                    # one 16-row band of the (18, 70)-residue pair needs 7.2 KB: no tiling -> CSR slabs + gather kernel
    {"MPCGPU_RELAX_LDS_KB": "12", "MPCGPU_RELAX_SHAPE": "2,2,5"},   # target 5 KB of 9.5: most steps prefetched, the large ones late
    {"MPCGPU_RELAX_TILES": "pairs"},                                 # the whole-record tiles of relax_var_kernel (A/B)
    {"MPCGPU_RELAX_FORM": "walk"},                                   # the two-list walk on block records for the Y operand too (no window records)
    {"MPCGPU_RELAX_FORM": "walk", "MPCGPU_RELAX_SHAPE": "4,2", "MPCGPU_RELAX_SLOTS": "2"},
    {"MPCGPU_RELAX_WIN_PCT": "100000"},                              # window records whatever they cost (the unrelated 70-residue sequence has wide rows)
    {"MPCGPU_RELAX_WIN_PCT": "100000", "MPCGPU_RELAX_SHAPE": "2,2,8", "MPCGPU_RELAX_LDS_KB": "20"},
    {"MPCGPU_RELAX_WIN_PCT": "100000", "MPCGPU_RELAX_LDS_KB": "9"},  # no band fits with windows: the store drops them and walks block lists
    {"MPCGPU_RELAX_ORDER": "pairs"},                                 # cells of an X group pair after pair (the default: blocks of 8 rows, a block's cells pair after pair)
    {"MPCGPU_RELAX_ORDER": "1"},                                     # row by row
    {"MPCGPU_RELAX_ORDER": "3", "MPCGPU_RELAX_SHAPE": "8,8", "MPCGPU_RELAX_SLOTS": "1"},  # blocks that do not divide the band
    {"MPCGPU_RELAX_ORDER": "pairs", "MPCGPU_RELAX_FORM": "walk", "MPCGPU_RELAX_SHAPE": "4,2", "MPCGPU_RELAX_SLOTS": "2"},
])
def test_emu_relax_band_tiles(emu, env):
    """relax_band_kernel over forced tile shapes, slot budgets and staging areas: row bands of 16..80 rows, Y ranges that start
    inside a record, ragged lengths (a 3-residue and a 70-residue sequence among 18-residue ones), both staging modes (next
    step prefetched beside the current one / staged after the merges), tiles split by the host — bit-identical to the oracle."""
    seqs = make_family(7, 75, seed=11) + make_family(3, 18, seed=5) + [make_family(1, 70, seed=9)[0], "MKV"]
    info = {}
    got = _with_env(env, lambda: P.run_lib(seqs, lib_path=emu, info=info))
    want = P.run_oracle(seqs)
    P.assert_same(got, want, "band tiles %s" % env)
    # (rows wider than the 5-bit span field of a window descriptor — the escape of the direct-index merge — are in this set: the
    # 3-residue sequence against a 75-residue one)
    assert max(int(v[1::2][o[i + 1] - 1]) - int(v[1::2][o[i]]) + 1 for o, v in want[0][0] for i in range(len(o) - 1) if o[i + 1] > o[i]) > 31
    if env.get("MPCGPU_RELAX_LDS_KB") == "9":
        assert "relax_band_kernel" in info["relax_info"] and "MpcRbBlocks" in info["relax_info"], info["relax_info"]
    elif "MPCGPU_RELAX_WIN_PCT" in env:
        assert "MpcRbWin" in info["relax_info"] and "window records" in info["relax_info"], info["relax_info"]
    elif env.get("MPCGPU_RELAX_FORM") == "walk":
        assert "MpcRbBlocks" in info["relax_info"] and "window records" not in info["relax_info"], info["relax_info"]
    if env.get("MPCGPU_RELAX_TILES") == "pairs":
        assert "relax_var_kernel" in info["relax_info"], info["relax_info"]
    elif env.get("MPCGPU_RELAX_LDS_KB") == "8":
        assert "relax_kernel" in info["relax_info"] and info["relax_fallback"], info["relax_info"]
    else:
        assert "relax_band_kernel" in info["relax_info"] and "band tiles" in info["relax_info"], info["relax_info"]


@pytest.mark.parametrize("mode", ["late", "reverse", "random"])
def test_emu_relax_band_tiles_races(emu, mode):
    """the band kernel's staging under the emulator's race finders: EMU_DMA=late (transfers land at the issuing thread's wait, the
    destination poisoned meanwhile) with step Z+1 prefetched beside step Z, and threads run in reverse / random order between
    synchronisation points"""
    seqs = make_family(6, 40, seed=3) + make_family(3, 18, seed=5)
    want = P.run_oracle(seqs)
    # the direct-index merge / the two-list walk / the row-by-row cell order (its tables are built in the staging area before the walk and
    # again after it: between the last step's readers and the epilogue's searches)
    for form in ({"MPCGPU_RELAX_WIN_PCT": "100000"}, {"MPCGPU_RELAX_FORM": "walk"}, {"MPCGPU_RELAX_WIN_PCT": "100000", "MPCGPU_RELAX_ORDER": "1"}):
        env = {"MPCGPU_RELAX_SHAPE": "4,4,3", "MPCGPU_RELAX_LDS_KB": "10"}
        env.update(form)
        env.update({"EMU_DMA": "late"} if mode == "late" else {"EMU_SCHED": mode})
        got = _with_env(env, lambda: P.run_lib(seqs, lib_path=emu))
        P.assert_same(got, want, "band tiles, %s, %s" % (mode, form))
