"""diag/rank_time.py — what ONE rank of a pair-sharded run does, timed on one GPU: rank 0's shard of `world` (stage A on its
pairs, import of all shards + store build, relax on its pairs, commit), everything except the two exchanges. The other ranks'
shards are computed beforehand on the same GPU (untimed) so that the imported store is the real one. With the 1-GPU step this
gives the compute side of the scaling curve the 1-GPU box cannot measure.   usage: python diag/rank_time.py [world] [N] [L] [rank]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from muscle_amd._lib import MpcGpu  # noqa: E402
from muscle_amd.mpcflat import shard_bounds  # noqa: E402
from muscle_amd.synth import make_family  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 400
me = int(sys.argv[4]) if len(sys.argv) > 4 else 0
seqs = make_family(n, L, seed=1)
lens = [len(s) for s in seqs]
cuts = shard_bounds(lens, world)
g = MpcGpu(0)
g.set_hmm(*bench.load_hmm())
if os.environ.get("RANK_TIME_NO_TIMERS"):
    g.timers_enable(False)
g.set_seqs(seqs)


def sync():
    g.synchronize()
    torch.cuda.synchronize()


# the other ranks' shards (untimed), this rank last so that its shard is the resident one
shards, sizes = [None] * world, [0] * world
for r in [q for q in range(world) if q != me] + [me]:
    g.calc_posteriors(cuts[r], cuts[r + 1])
    nb, _ = g.shard_info()
    t = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
    g.shard_export(t.data_ptr())
    shards[r], sizes[r] = t, nb
sync()
res = {}
for rep in range(2):
    t0 = time.perf_counter()
    g.calc_posteriors(cuts[me], cuts[me + 1])
    sync()
    t1 = time.perf_counter()
    nb, _ = g.shard_info()
    mine = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
    g.shard_export(mine.data_ptr())
    full = torch.cat(shards[:me] + [mine] + shards[me + 1:])  # stands in for the all-gather (device-local copy)
    sync()
    t2 = time.perf_counter()
    g.store_import(cuts[:-1], cuts[1:], sizes, full.data_ptr())
    sync()
    t3 = time.perf_counter()
    relax = commit = 0.0
    res_it = []
    for _ in range(2):
        a = time.perf_counter()
        g.cons_iter(cuts[me], cuts[me + 1])
        sync()
        b = time.perf_counter()
        # the other ranks' values would arrive here; committing rank 0's own values into the whole store costs the same
        g.cons_commit()
        sync()
        c = time.perf_counter()
        relax += b - a
        commit += c - b
        res_it = res_it + [b - a] if _ else [b - a]
    res = {"stage_a": t1 - t0, "export+concat": t2 - t1, "import+store": t3 - t2, "relax": relax, "relax it.1": res_it[0], "relax it.2": res_it[1], "commit": commit}
tot = sum(v for k, v in res.items() if not k.startswith("relax it"))
print("world %d, rank %d of %d x L~%d (%d of %d pairs): " % (world, me, n, L, cuts[me + 1] - cuts[me], cuts[-1]) +
      ", ".join("%s %.1f ms" % (k, 1e3 * v) for k, v in res.items()) + "; total %.1f ms without the two exchanges" % (1e3 * tot))
