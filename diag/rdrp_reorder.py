"""diag/rdrp_reorder.py — experiment for DESIGN.md 4.3 (round-5 review, item 4a): does grouping SIMILAR sequences into the same band tiles
(tiles take 8 x 8 consecutive sequence indices) make the real-data relax cheaper? Reorders the first N rdrp records by the leaf order of an
average-linkage tree over the stage's own EA values and writes a FASTA the bench can take with --fasta; relax time and tile geometry of the two
orders are compared (the reordered run's posteriors are NOT the original's: the sum order over Z changes — this measures the lever only).
usage: python diag/rdrp_reorder.py [N] [out.fa]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from muscle_amd._lib import MpcGpu  # noqa: E402
from muscle_amd.synth import read_fasta  # noqa: E402
from scipy.cluster.hierarchy import leaves_list, linkage  # noqa: E402
from scipy.spatial.distance import squareform  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/rdrp_reordered.fa"
seqs = read_fasta(os.path.join(ROOT, "tests", "golden", "rdrp_first1000.fa.gz"))[:n]
g = MpcGpu(0)
g.set_hmm(*bench.load_hmm())
g.set_seqs(seqs)
g.calc_posteriors()
ea = g.get_ea().astype(np.float64)
g.close()
d = np.clip(1.0 - ea, 0.0, None)  # upgma5.cpp:504-519: the tree's distance is 1 - EA
order = leaves_list(linkage(d, method="average"))  # condensed form == InitPairs order
with open(out, "w") as f:
    for k, i in enumerate(order):
        f.write(">s%d_orig%d\n%s\n" % (k, int(i), seqs[int(i)]))
D = squareform(d)
adj = float(np.mean([D[order[k], order[k + 1]] for k in range(n - 1)]))
print("wrote %s: %d records in tree-leaf order; mean distance of index neighbours %.3f (original order %.3f, all pairs %.3f)"
      % (out, n, adj, float(np.mean([D[k, k + 1] for k in range(n - 1)])), float(d.mean())))
