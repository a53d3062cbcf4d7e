"""Pins the CPU oracle against the committed golden fixtures (outputs of the compiled reference,
tests/golden/make_golden.py). Runs anywhere — this is the oracle's pin on the GPU box."""
import numpy as np
import pytest

import _golden as G
import _oracle as O


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def hmm():
    s, t, m, i, thr = G.hmm_tables()
    assert bits(O.lib().orc_min_sparse_score()) == bits(thr)
    return O.make_hmm(s, t, m, i)


def test_pairs_small(hmm):
    z = G.load("pairs_small")
    for k in range(int(z["n"])):
        x, y = z["x%d" % k].tobytes(), z["y%d" % k].tobytes()
        LX, LY = len(x), len(y)
        F, B = O.fwd(hmm, x, y), O.bwd(hmm, x, y)
        assert G.sha(F) == str(z["Fsha%d" % k]) and G.sha(B) == str(z["Bsha%d" % k])
        assert np.array_equal(bits(F.reshape(LX + 1, LY + 1, 5)[:, :, 0]), bits(z["FM%d" % k]))
        assert bits(O.total(F, B, LX, LY)) == bits(z["total%d" % k])
        P = O.post(F, B, LX, LY)
        assert np.array_equal(bits(P), bits(z["post%d" % k]))
        off, val = O.sparse_from_post(P)
        assert np.array_equal(off, z["off%d" % k]) and np.array_equal(val, z["val%d" % k])
        assert bits(O.aln_score(P)) == bits(z["alnscore%d" % k])
        sc, path = O.calc_aln(P)
        assert path == str(z["path%d" % k]) and bits(sc) == bits(z["calcaln_score%d" % k])


@pytest.mark.parametrize("name", G.MPC_SETS)
def test_mpc_stage(hmm, name):
    g = G.mpc(name)
    st = O.Store(g["seqs"])
    ea = st.calc_posteriors(hmm)
    assert np.array_equal(bits(ea), bits(g["ea"]))
    cur = st
    for s in range(g["nstages"]):
        if s > 0:
            cur = cur.cons_iter()
        stage = [cur.get(k) for k in range(st.npairs)]
        assert G.stage_digest(stage) == g["digest"][s], "stage %d" % s
        if g["stage"][s] is not None:
            for (o1, v1), (o2, v2) in zip(stage, g["stage"][s]):
                assert np.array_equal(o1, o2) and np.array_equal(v1, v2)


@pytest.mark.parametrize("name", G.MEGA_SETS)
def test_mega_stage(hmm, name):
    """Structure-profile emissions (calcpost.cpp:14-22 -> fwdflat_mega.cpp / bwdflat_mega.cpp / mega.cpp:273-363):
    the oracle on the reference's parsed tables vs the reference's own outputs."""
    m = G.mega(name)
    z = m["z"]
    g = O.make_mega(m["alpha"], m["weight"], m["lp"], m["mx"])
    p0, p1 = m["profs"][0], m["profs"][1]
    l0, l1 = len(m["seqs"][0]), len(m["seqs"][1])
    ins0 = np.array([O.mega_ins(g, p0, p) for p in range(l0)], np.float32)
    assert np.array_equal(bits(ins0), bits(z["ins0"]))
    mt = np.array([[O.mega_match(g, p0, a, p1, b) for b in range(l1)] for a in range(l0)], np.float32)
    assert np.array_equal(bits(mt), bits(z["match01"]))
    F, B = O.fwd_mega(hmm, g, p0, p1), O.bwd_mega(hmm, g, p0, p1)
    assert G.sha(F) == str(z["F01_sha"]) and G.sha(B) == str(z["B01_sha"])
    assert np.array_equal(bits(F.reshape(l0 + 1, l1 + 1, 5)[:, :, 0]), bits(z["F01_M"]))
    st = O.MegaStore(m["seqs"], m["profs"])
    ea = st.calc_posteriors_mega(hmm, g)
    assert np.array_equal(bits(ea), bits(m["ea"]))
    cur = st
    for s in range(m["nstages"]):
        if s > 0:
            cur = cur.cons_iter()
        stage = [cur.get(k) for k in range(st.npairs)]
        assert G.stage_digest(stage) == m["digest"][s], "stage %d" % s
        for (o1, v1), (o2, v2) in zip(stage, m["stage"][s]):
            assert np.array_equal(o1, o2) and np.array_equal(v1, v2)


def test_expf_emulation_matches_libm():
    """The glibc-2.35 expf restatement (both ifunc variants) vs this host's libm over the only
    range the path uses, [logf(0.01f), 0): the variant the host resolves to must be bit-identical."""
    L = O.lib()
    use_fma = L.orc_host_expf_uses_fma()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-4.7, 0.0, 200000).astype(np.float32),
                         np.float32([-4.6051702, -4.605170, -1e-7, -0.5, -1, -2, -3, -4])])
    bad = 0
    for x in xs:
        if bits(L.orc_expf_emul(float(x), use_fma)) != bits(L.orc_libm_expf(float(x))):
            bad += 1
    assert bad == 0, "%d mismatches (use_fma=%d)" % (bad, use_fma)


def test_oracle_vs_config2_block0(hmm):
    """BASELINE config 2 (256 x L~300) is pinned by reference-generated digests (mpcbig_n256_L300.npz). The oracle
    reproduces the first two 1000-pair blocks of stage 0 and their EA values (sequences 0..3 against all later ones) —
    a size the CPU suite can afford; the whole set is the GPU suite's."""
    import hashlib
    import _bigdigest as D
    from muscle_amd.synth import make_family
    name = "n256_L300"
    if D.fixture_for(*D.BIG_SETS[name]) is None:
        pytest.skip("fixture not generated")
    z = D.load(name)
    n, length, seed = D.BIG_SETS[name]
    seqs = make_family(n, length, seed=seed)
    assert hashlib.sha256("\n".join(seqs).encode()).hexdigest() == str(z["seqs_sha"])
    st = O.Store(seqs)
    ea = st.calc_posteriors(hmm, 0, 2000, threads=0)
    block = int(z["block"])
    for b in range(2):
        h = hashlib.sha256()
        for k in range(b * block, (b + 1) * block):
            off, val = st.get(k)
            h.update(np.ascontiguousarray(off, np.uint32).tobytes())
            h.update(np.ascontiguousarray(val, np.uint32).tobytes())
        assert h.digest() == z["blocks0"][b].tobytes(), "block %d" % b
        assert hashlib.sha256(np.ascontiguousarray(ea[b * block:(b + 1) * block], np.float32).tobytes()).digest() == z["ea_blocks"][b].tobytes()


@pytest.mark.parametrize("name", ["bp_n9_L40"])
def test_buildpost_restatement_vs_reference_golden(name):
    """tests/_buildpost.py (numpy restatement of buildpostflat.cpp:18-106 the older device tests compare with) on the ORACLE's
    store after two relax iterations reproduces the compiled reference's matrices bit for bit: pins the oracle's relax and the
    restatement together."""
    import _buildpost as BP
    import _parity as P
    z = G.load(name)
    seqs = [str(x) for x in z["seqs"]]
    stages, _ea = P.run_oracle(seqs)
    n = len(seqs)
    pidx = {p: k for k, p in enumerate((a, b) for a in range(n) for b in range(a + 1, n))}
    for j in range(int(z["njoins"])):
        k = "j%d_" % j
        grp1, grp2 = [int(x) for x in z[k + "idx1"]], [int(x) for x in z[k + "idx2"]]
        rows1, rows2 = [str(x) for x in z[k + "rows1"]], [str(x) for x in z[k + "rows2"]]
        m1, m2 = [BP.pos_to_col(r) for r in rows1], [BP.pos_to_col(r) for r in rows2]
        got = BP.build_post(stages[2], pidx, grp1, grp2, m1, m2, len(rows1[0]), len(rows2[0]))
        assert np.array_equal(P.bits(got), P.bits(z[k + "post"])), j
        w = z[k + "w"]
        gotw = BP.build_post(stages[2], pidx, grp1, grp2, m1, m2, len(rows1[0]), len(rows2[0]), w[:len(grp1)], w[:len(grp2)])
        assert np.array_equal(P.bits(gotw), P.bits(z[k + "postw"])), ("weighted", j)
