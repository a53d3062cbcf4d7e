"""diag/rank_time.py — what ONE rank of a pair-sharded run does, timed on one GPU: the rank's blocks of `world` (stage A on its
pairs in `pieces` pieces, import of all shards + PARTIAL store build, relax on its pairs, commit), everything except the two
exchanges. The other ranks' shards are computed beforehand on the same GPU (untimed) so that the imported store is the real one. With
the 1-GPU step this gives the compute side of the scaling curve the 1-GPU box cannot measure.
usage: python diag/rank_time.py [world] [N] [L] [rank] [pieces]"""
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
from muscle_amd.mpcflat import parse_pieces, piece_cuts, plan  # noqa: E402
from muscle_amd.synth import make_family  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 400
me = int(sys.argv[4]) if len(sys.argv) > 4 else 0
SPEC = sys.argv[5] if len(sys.argv) > 5 else "1"  # pieces: a count, or shares of the DP cells ("0.85,0.15")
P = len(parse_pieces(SPEC))
seqs = make_family(n, L, seed=1)
lens = [len(s) for s in seqs]
g = MpcGpu(0)
g.set_hmm(*bench.load_hmm())
if os.environ.get("RANK_TIME_NO_TIMERS"):
    g.timers_enable(False)
g.set_seqs(seqs)
rects, pos, px, py = plan(g, lens, world)
g.set_pair_order(rects)
cuts = piece_cuts(lens, px, py, pos, SPEC)
need = len(set(px[pos[me]:pos[me + 1]].tolist()) | set(py[pos[me]:pos[me + 1]].tolist()))


def sync():
    g.synchronize()
    torch.cuda.synchronize()


def pad16(x):
    return (x + 15) & ~15


# the other ranks' shards (untimed), one shard per rank
others = {}
for r in [q for q in range(world) if q != me]:
    g.calc_posteriors(pos[r], pos[r + 1])
    nb, _ = g.shard_info()
    t = torch.empty(pad16(nb), dtype=torch.uint8, device="cuda:0")
    g.shard_export(t.data_ptr())
    others[r] = (t, nb)
sync()
res = {}
for rep in range(2):
    g.timers_reset()
    t0 = time.perf_counter()
    mine = []
    for p in range(P):
        g.calc_posteriors(cuts[me][p], cuts[me][p + 1])
        nb, _ = g.shard_info()
        t = torch.empty(pad16(nb), dtype=torch.uint8, device="cuda:0")
        g.shard_export(t.data_ptr())  # (stands in for the copy into the gather buffer)
        mine.append((t, nb))
    sync()
    t1 = time.perf_counter()
    parts, k0s, k1s, sizes, offs, at = [], [], [], [], [], 0
    for r in range(world):
        for (t, nb), a, b in ([(others[r], pos[r], pos[r + 1])] if r != me else [(mine[p], cuts[me][p], cuts[me][p + 1]) for p in range(P)]):
            parts.append(t); k0s.append(a); k1s.append(b); sizes.append(nb); offs.append(at)
            at += t.numel()
    full = torch.cat(parts)  # stands in for the all-gather (device-local copy)
    sync()
    t2 = time.perf_counter()
    g.store_import_part(k0s, k1s, sizes, offs, full.data_ptr(), pos[me], pos[me + 1])
    sync()
    t3 = time.perf_counter()
    relax = commit = 0.0
    res_it = []
    for it in range(2):
        a = time.perf_counter()
        g.cons_iter(pos[me], pos[me + 1])
        sync()
        b = time.perf_counter()
        # the other ranks' values would arrive here; committing the rank's own values into the whole store costs the same
        g.cons_commit()
        sync()
        c = time.perf_counter()
        relax += b - a
        commit += c - b
        res_it.append(b - a)
    res = {"stage_a": t1 - t0, "export+concat": t2 - t1, "import+store": t3 - t2, "relax": relax, "relax it.1": res_it[0], "relax it.2": res_it[1], "commit": commit}
    kern = {k: round(v[0], 2) for k, v in g.timers_get().items() if v[0]}
tot = sum(v for k, v in res.items() if not k.startswith("relax it") and k != "export+concat")
print("world %d, rank %d of %d x L~%d (%d of %d pairs, %d pieces, %d rectangles; the rank's store holds %d of %d sequences): " % (world, me, n, L, pos[me + 1] - pos[me], pos[-1], P, len(rects), need, n) +
      ", ".join("%s %.1f ms" % (k, 1e3 * v) for k, v in res.items()) + "; total %.1f ms without the two exchanges (and without the stand-in copies); kernels %s; %s" % (1e3 * tot, kern, g.relax_info()[0]))
