"""diag/fuzz_parity.py — seeded random inputs through the library and the oracle for a time budget, every stage compared bit for bit
(tests/_parity.py: EA, stage-0 sparse posteriors, the matrices after each relax iteration; then random joins: BuildPost + CalcAlnFlat
against the numpy restatement + the oracle's DP). The committed tests pin chosen shapes; this walks the space between them: 2..48
sequences, lengths 1..3000, unrelated / identical / low-complexity sequences, mixtures of families, real proteins (rdrp picks),
nucleotides, arbitrary seven-bit letters. A failure prints the case's seed (rerun: python diag/fuzz_parity.py 0 <seed>) and the run
goes on; exit status 1 if any case failed. TEST INFRASTRUCTURE (imports the oracle).
With `group`, every case also runs as a GROUP of 2..8 contexts on device 0 (mpcgpu_group_*: the block partition, partial stores, the two
exchanges by peer copies) and every rank's stages are compared as well.
With `medium`: 65..200 sequences per case (families, ragged families, rdrp picks) — past the tile cutter's shortcut for few sequences.
usage: python diag/fuzz_parity.py [seconds] [first seed] [group | medium | medium,group]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _buildpost as BP  # noqa: E402
import _golden as G  # noqa: E402
import _oracle as O  # noqa: E402
import _parity as P  # noqa: E402
from muscle_amd._lib import MpcGpu, MpcGroup  # noqa: E402
from muscle_amd.synth import AMINO, make_family, read_fasta  # noqa: E402

MEDIUM = False  # main(): the `medium` mode
RDRP = read_fasta(os.path.join(ROOT, "tests", "golden", "rdrp_first1000.fa.gz"))


def make_case(seed):
    """-> (description, sequences, hmm name, relax iterations)"""
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 10))
    iters = int(rng.integers(1, 4))
    hmm = "hmm_amino"
    if MEDIUM:  # more than 64 sequences: the tile cutter's search over shapes and targets instead of its few-sequences shortcut
        n = int(rng.integers(65, 200))
        if kind < 3:
            seqs = [RDRP[i] for i in rng.choice(len(RDRP), size=min(n, 130), replace=False)]
            return "medium rdrp n=%d" % len(seqs), seqs, hmm, min(iters, 2)
        L = int(rng.integers(30, 260))
        sub = float(rng.choice([0.1, 0.3, 0.6]))
        seqs = make_family(n, L, seed=seed, p_del=float(rng.uniform(0, 0.1)), p_ins=float(rng.uniform(0, 0.1)), p_sub=sub)
        if kind >= 7:  # ragged: a third of them fragments
            for i in range(0, n, 3):
                a, b = sorted(int(x) for x in rng.integers(0, len(seqs[i]) + 1, size=2))
                seqs[i] = seqs[i][a:b] if b > a else seqs[i][:1]
        return "medium family n=%d L=%d sub=%.1f%s" % (n, L, sub, " ragged" if kind >= 7 else ""), seqs, hmm, min(iters, 2)
    if kind == 0:  # a family, any divergence
        n, L = int(rng.integers(2, 49)), int(rng.integers(1, 400))
        sub = float(rng.choice([0.0, 0.02, 0.3, 0.6, 0.95]))
        seqs = make_family(n, L, seed=seed, p_del=float(rng.uniform(0, 0.2)), p_ins=float(rng.uniform(0, 0.2)), p_sub=sub)
        what = "family n=%d L=%d sub=%.2f" % (n, L, sub)
    elif kind == 1:  # tiny sequences
        n = int(rng.integers(2, 30))
        seqs = ["".join(rng.choice(list(AMINO), size=int(rng.integers(1, 6)))) for _ in range(n)]
        what = "tiny n=%d" % n
    elif kind == 2:  # one or two long ones among short ones
        n, L = int(rng.integers(3, 12)), int(rng.integers(800, 3000))
        seqs = make_family(n - 2, int(rng.integers(5, 120)), seed=seed) + make_family(2, L, seed=seed + 1)
        order = rng.permutation(n)
        seqs = [seqs[i] for i in order]
        what = "long n=%d L=%d" % (n, L)
    elif kind == 3:  # low complexity: wide rows
        n, L = int(rng.integers(3, 24)), int(rng.integers(10, 200))
        letters = list(rng.choice(list(AMINO), size=int(rng.integers(1, 4))))
        seqs = ["".join(rng.choice(letters, size=max(1, L + int(rng.integers(-8, 9))))) for _ in range(n)]
        what = "low complexity n=%d L=%d letters=%d" % (n, L, len(letters))
    elif kind == 4:  # two or three unrelated families
        parts = [make_family(int(rng.integers(1, 12)), int(rng.integers(20, 300)), seed=seed + 7 * f) for f in range(int(rng.integers(2, 4)))]
        seqs = [s for p in parts for s in p]
        seqs = [seqs[i] for i in rng.permutation(len(seqs))]
        what = "mixture n=%d" % len(seqs)
    elif kind == 5:  # real proteins
        n = int(rng.integers(3, 28))
        seqs = [RDRP[i] for i in rng.choice(len(RDRP), size=n, replace=False)]
        what = "rdrp n=%d" % n
    elif kind == 6:  # nucleotides
        n, L = int(rng.integers(2, 40)), int(rng.integers(1, 500))
        fam = make_family(n, L, seed=seed, p_sub=float(rng.choice([0.05, 0.3])))
        seqs = ["".join("ACGT"[AMINO.index(c) % 4] for c in s) for s in fam]
        hmm = "hmm_nucleo"
        what = "nucleotides n=%d L=%d" % (n, L)
    elif kind == 7:  # any seven-bit letters
        n, L = int(rng.integers(2, 16)), int(rng.integers(1, 160))
        base = rng.integers(1, 128, size=L)
        seqs = []
        for _ in range(n):
            s = base.copy()
            flip = rng.random(L) < 0.3
            s[flip] = rng.integers(1, 128, size=int(flip.sum()))
            seqs.append(bytes(s.astype(np.uint8)))
        what = "bytes n=%d L=%d" % (n, L)
    elif kind == 8:  # identical sequences, and duplicates among others
        n, L = int(rng.integers(2, 20)), int(rng.integers(1, 250))
        one = make_family(1, L, seed=seed)[0]
        seqs = [one] * n + make_family(int(rng.integers(0, 5)), L, seed=seed + 3)
        what = "identical n=%d L=%d" % (len(seqs), L)
    else:  # skewed lengths of one family: prefixes and suffixes
        n, L = int(rng.integers(3, 32)), int(rng.integers(30, 600))
        fam = make_family(n, L, seed=seed)
        seqs = []
        for s in fam:
            a, b = sorted(int(x) for x in rng.integers(0, len(s) + 1, size=2))
            seqs.append(s[a:b] if b > a else s[:1])
        what = "fragments n=%d L=%d" % (n, L)
    return what, seqs, hmm, iters


def mega_for(what, seqs, hmm, seed):
    """structure profiles (Mega emissions: fwdflat_mega.cpp, bwdflat_mega.cpp) for a fifth of the amino-acid cases of str sequences"""
    if hmm != "hmm_amino" or isinstance(seqs[0], bytes) or seed % 5 != 0 or any(c not in AMINO for s in seqs for c in s):
        return None
    return P.random_mega(seqs, seed, nfeat=int(np.random.default_rng(seed + 4).integers(2, 9)))


def check_joins(seqs, hmm, seed):
    """random bipartitions of random subsets on random gapped rows: mpcgpu_align_alns against the restatement"""
    rng = np.random.default_rng(seed + 1)
    n = len(seqs)
    if n < 3 or n > 14 or max(len(s) for s in seqs) > 150 or isinstance(seqs[0], bytes):
        return 0
    s, t, m, i, thr = G.hmm_tables(hmm)
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
    done = 0
    for _ in range(3):
        order = [int(x) for x in rng.permutation(n)[:int(rng.integers(2, n + 1))]]
        cut = int(rng.integers(1, len(order)))
        grp1, grp2 = order[:cut], order[cut:]
        rows1, C1 = BP.random_msa(seqs, grp1, rng)
        rows2, C2 = BP.random_msa(seqs, grp2, rng)
        m1 = [BP.pos_to_col(r) for r in rows1]
        m2 = [BP.pos_to_col(r) for r in rows2]
        w1 = rng.uniform(0.2, 1.8, len(grp1)).astype(np.float32) if rng.random() < 0.5 else None
        w2 = rng.uniform(0.2, 1.8, len(grp2)).astype(np.float32) if w1 is not None else None
        sc0, path0 = O.calc_aln(BP.build_post(stage, pidx, grp1, grp2, m1, m2, C1, C2, w1, w2))
        path, sc = g.align_alns(grp1, grp2, m1, m2, C1, C2, w1, w2) if w1 is not None else g.align_alns(grp1, grp2, m1, m2, C1, C2)
        assert path == path0 and P.bits(sc) == P.bits(sc0), ("join", grp1, grp2)
        done += 1
    g.close()
    return done


def check_group(seqs, hmm, iters, want, seed):
    """the same case on a group of contexts: every rank's EA and stages against the oracle's"""
    rng = np.random.default_rng(seed + 2)
    world = int(rng.choice([2, 3, 4, 5, 8]))
    if len(seqs) < 3 or max(len(s) for s in seqs) > 1200:
        return 0
    grp = MpcGroup([0] * world)
    grp.set_hmm(*G.hmm_tables(hmm))
    grp.set_seqs(seqs)
    grp.calc_posteriors()
    views = [grp.ctx(r) for r in range(world)]
    stages_w, ea_w = want
    for r, v in enumerate(views):
        P.assert_same(([v.get_sparse_range()], v.get_ea()), ([stages_w[0]], ea_w), "group of %d, rank %d, stage 0" % (world, r))
    for it in range(iters):
        grp.cons_iter()
        for r, v in enumerate(views):
            P.assert_same(([v.get_sparse_range()], ea_w), ([stages_w[it + 1]], ea_w), "group of %d, rank %d, stage %d" % (world, r, it + 1))
    del views
    grp.close()
    return 1


def main():
    BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    GROUP = len(sys.argv) > 3 and "group" in sys.argv[3].split(",")
    global MEDIUM
    MEDIUM = len(sys.argv) > 3 and "medium" in sys.argv[3].split(",")
    t0 = time.time()
    seed, cases, joins, groups, failed = SEED0, 0, 0, 0, []
    kinds = {}
    while (time.time() - t0 < BUDGET) if BUDGET > 0 else (seed == SEED0):
        what, seqs, hmm, iters = make_case(seed)
        try:
            mega = mega_for(what, seqs, hmm, seed)
            if mega is not None:
                what = "mega " + what
            got = P.run_lib(seqs, iters=iters, hmm_name=hmm, mega=mega)
            want = P.run_oracle(seqs, iters=iters, hmm_name=hmm, mega=mega)
            P.assert_same(got, want, what)
            if mega is not None:
                pass
            elif GROUP:
                groups += check_group(seqs, hmm, iters, want, seed)
            else:
                joins += check_joins(seqs, hmm, seed)
            kinds[what.split()[0]] = kinds.get(what.split()[0], 0) + 1
        except Exception:  # noqa: BLE001 — report and go on
            failed.append((seed, what))
            print("FAILED seed %d: %s" % (seed, what), flush=True)
            traceback.print_exc()
        cases += 1
        seed += 1
    print("fuzz_parity: %d cases (%s), %d joins, %d groups of contexts, seeds %d..%d, %.0f s: %s" % (cases, ", ".join("%s %d" % kv for kv in sorted(kinds.items())), joins, groups,
          SEED0, seed - 1, time.time() - t0, "all bit-identical to the oracle" if not failed else "%d FAILED: %s" % (len(failed), failed)), flush=True)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
