"""diag/join_bench.py — throughput of the PProg join entry point (mpcgpu_align_msas: stage A on an explicit list of
cross pairs + CalcPosteriorFlat3 + CalcAlnFlat) on synthetic data. usage: python diag/join_bench.py [NSEQ LEN PAIRS REPS]
MPCGPU_LIB overrides the library (emulator dry run). Prints one line per repetition and the best rate."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muscle_amd._lib import MpcGpu  # noqa: E402
from muscle_amd.synth import make_family  # noqa: E402

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 400
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
npairs = int(sys.argv[3]) if len(sys.argv) > 3 else 2000  # getpairs.cpp:33-69 samples at most 2000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
seqs = make_family(nseq, L, seed=3)
half = nseq // 2
rng = np.random.default_rng(1)
s1 = rng.integers(0, half, npairs).astype(np.uint32)
s2 = rng.integers(half, nseq, npairs).astype(np.uint32)
# the two "MSAs" are the ungapped sequences left-aligned: position p sits in column p
C1 = max(len(seqs[i]) for i in range(half))
C2 = max(len(seqs[i]) for i in range(half, nseq))
p2c1 = [np.arange(len(seqs[i]), dtype=np.uint32) for i in s1]
p2c2 = [np.arange(len(seqs[i]), dtype=np.uint32) for i in s2]
z = np.load(os.path.join(ROOT, "tests", "golden", "hmm_amino.npz"))
g = MpcGpu(0, os.environ.get("MPCGPU_LIB") or None)
g.set_hmm(z["start"], z["trans"], z["match"], z["ins"], np.float32(z["min_sparse_score"]))
g.set_seqs_registry(seqs)
best = 0.0
for r in range(reps):
    t0 = time.perf_counter()
    path, score, ea = g.align_msas(s1, s2, p2c1, p2c2, C1, C2)
    dt = time.perf_counter() - t0
    best = max(best, npairs / dt)
    print("rep %d: %d pairs of L~%d in %.3f s = %.0f pairs/s (path %d columns, score %.3f, mean EA %.4f)"
          % (r, npairs, L, dt, npairs / dt, len(path), score, float(ea.mean())), flush=True)
print("best: %.0f pairs/s" % best)
g.close()
