"""Digest-only parity check of a BASELINE-size run (configs 2 and 3) against the reference-generated fixtures
tests/golden/mpcbig_<name>.npz (made by tests/golden/make_golden.py big ... from oracle/_ref/libmuscle_ref.so =
MPCFlat::CalcPosteriors + ConsIter x2 of the compiled reference): sha256 per block of 1000 pairs over
(offsets u32[LX+1] || values {P bits, col} u32[2 nnz]) per pair in InitPairs order, the whole-stage sha256, and the
EA values' digests. Used by the -m gpu tests and by bench.py's self-check after the timed region."""
import hashlib
import os

import numpy as np

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BIG_SETS = {"n256_L300": (256, 300, 1), "n1000_L400": (1000, 400, 1), "rdrp256": (256, 0, 0), "rdrp384": (384, 0, 0)}  # length 0: first n rdrp records


def fixture_for(n, length, seed):
    for name, key in BIG_SETS.items():
        if key == (n, length, seed) and os.path.exists(os.path.join(GDIR, "mpcbig_%s.npz" % name)):
            return name
    return None


def fixture_for_fasta(path, n):
    """first n records of tests/golden/rdrp_first1000.fa.gz -> fixture name, when the compiled reference has digested that prefix"""
    if os.path.basename(path) != "rdrp_first1000.fa.gz":
        return None
    return fixture_for(n, 0, 0)


def seqs_of(name):
    """the sequences of a big set (synthetic family or rdrp prefix)"""
    from muscle_amd.synth import make_family, read_fasta
    n, length, seed = BIG_SETS[name]
    if length == 0:
        return read_fasta(os.path.join(GDIR, "rdrp_first1000.fa.gz"))[:n]
    return make_family(n, length, seed=seed)


def load(name):
    return np.load(os.path.join(GDIR, "mpcbig_%s.npz" % name), allow_pickle=False)


def stage_digests(g, block=1000, k0=0, k1=None):
    """g: MpcGpu with a current store. -> (whole-stage hexdigest, [block digests as bytes], total nnz) of the
    current sparse matrices of pairs [k0,k1), downloaded block by block through mpcgpu_get_sparse_range."""
    k1 = g.npairs if k1 is None else k1
    whole, blocks, tot = hashlib.sha256(), [], 0
    for b0 in range(k0, k1, block):
        b1 = min(b0 + block, k1)
        blk = hashlib.sha256()
        for off, val in g.get_sparse_range(b0, b1):
            ob, vb = off.tobytes(), val.tobytes()
            whole.update(ob); whole.update(vb)
            blk.update(ob); blk.update(vb)
            tot += len(val) // 2
        blocks.append(blk.digest())
    return whole.hexdigest(), blocks, tot


def ea_digests(ea, block=1000):
    ea = np.ascontiguousarray(ea, np.float32)
    return hashlib.sha256(ea.tobytes()).hexdigest(), [hashlib.sha256(ea[b:b + block].tobytes()).digest()
                                                      for b in range(0, len(ea), block)]


def compare_stage(z, s, g):
    """-> None when stage s of fixture z equals the library's current store, else a short description."""
    block = int(z["block"])
    whole, blocks, tot = stage_digests(g, block)
    if whole == str(z["digest%d" % s]):
        return None
    want = z["blocks%d" % s]
    bad = [b for b in range(len(blocks)) if blocks[b] != want[b].tobytes()]
    return "stage %d: nnz %d (reference %d), %d of %d blocks of %d pairs differ, first = block %d" % (
        s, tot, int(z["nnz_total%d" % s]), len(bad), len(blocks), block, bad[0] if bad else -1)


def compare_ea(z, ea):
    whole, blocks = ea_digests(ea, int(z["block"]))
    if whole == str(z["ea_sha"]):
        return None
    want = z["ea_blocks"]
    bad = [b for b in range(len(blocks)) if blocks[b] != want[b].tobytes()]
    return "EA: %d of %d blocks differ, first = block %d" % (len(bad), len(blocks), bad[0] if bad else -1)


# ---- SAMPLED pin of the real-data bench size (rdrp, N = 1000): the compiled reference's stage A for ALL pairs (EA + stage-0 block
# digests) and MPCFlat::ConsPair of iteration 1 for 2048 seeded pairs (tests/golden/make_golden.py big-sampled; a full ConsIter at
# this size is days of RelaxFlat_XZ_YZ's linear search). A fixture of its own: mpcbig_rdrp<n>_sampled.npz.
def sampled_fixture_for_fasta(path, n):
    if os.path.basename(path) != "rdrp_first1000.fa.gz":
        return None
    name = "rdrp%d_sampled" % n
    return name if os.path.exists(os.path.join(GDIR, "mpcbig_%s.npz" % name)) else None


def compare_sample(z, g):
    """the library's CURRENT store (after one relax iteration + commit) against the sampled pairs' reference digests -> None | text"""
    import hashlib as H
    ks, want = z["sample_k"], z["sample_sha1"]
    bad = []
    for q, k in enumerate(ks):
        (off, val), = g.get_sparse_range(int(k), int(k) + 1)
        if H.sha256(off.tobytes() + val.tobytes()).digest() != want[q].tobytes():
            bad.append(int(k))
    return None if not bad else "stage 1 sample: %d of %d sampled pairs differ, first = pair %d" % (len(bad), len(ks), bad[0])


def stage2_fixture_for_fasta(path, n):
    """the stage-2 pin of a few pairs of the rdrp prefix (tests/golden/make_golden.py big-stage2), or None"""
    if os.path.basename(path) != "rdrp_first1000.fa.gz":
        return None
    name = "rdrp%d_stage2_clique" % n
    return name if os.path.exists(os.path.join(GDIR, "mpcbig_%s.npz" % name)) else None


def compare_stage2_clique(z, g):
    """the library's CURRENT store (after TWO relax iterations + commits) against the compiled reference's stage-2 matrices of the
    pairs among the fixture's clique of sequences -> None | text"""
    import hashlib as H
    ks, want = z["stage2_k"], z["stage2_sha"]
    bad = []
    for q, k in enumerate(ks):
        (off, val), = g.get_sparse_range(int(k), int(k) + 1)
        if H.sha256(off.tobytes() + val.tobytes()).digest() != want[q].tobytes():
            bad.append(int(k))
    return None if not bad else "stage 2 clique: %d of %d pairs differ, first = pair %d" % (len(bad), len(ks), bad[0])
