"""Host-side mirror of the slice of class MPCFlat (mpcflat.h:16-105) that the hot path touches,
for tests and bench.py: InitSeqs/InitPairs -> CalcPosteriors -> Consistency, single GPU or
pair-sharded over torch.distributed ranks (one process per GPU; RCCL all-gather of the packed
sparse posteriors, then of the relaxed values — SURVEY.md §8e). Compute is ONLY the C-ABI library;
the `engine` argument lets the gloo CPU tests substitute a recording stand-in to check the sharding
and exchange logic without a GPU (there is no CPU compute path in the product).
"""
import numpy as np

CONSISTENCY_ITERS = 2  # DEFAULT_CONSISTENCY_ITERS_FLAT, mpcflat.h:12


def pair_lengths(lens):
    """(LX, LY) arrays over pairs in InitPairs order (mpcflat.cpp:139-159)."""
    lens = np.asarray(lens, np.int64)
    n = len(lens)
    ii, jj = np.triu_indices(n, 1)
    return lens[ii], lens[jj]


def shard_bounds(lens, world):
    """Contiguous pair ranges balanced by DP cells sum (LX+1)(LY+1) (SURVEY.md §8e partitioning).
    Returns world+1 cut points; deterministic, identical on every rank."""
    lx, ly = pair_lengths(lens)
    w = np.cumsum((lx + 1) * (ly + 1))
    total = int(w[-1]) if len(w) else 0
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        cuts.append(int(np.searchsorted(w, target, side="left")))
    cuts.append(len(lx))
    for r in range(1, len(cuts)):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts


class TorchExchange:
    """The two collectives of the sharded stage, over torch.distributed (backend nccl == RCCL on
    ROCm; gloo in the CPU tests). Shards differ in size, so the all-gather is `world` broadcasts,
    each of one rank's exact segment straight into its final place in ONE preallocated buffer
    (queued together, waited for once): no padding to the largest shard and no concatenation copy."""

    def __init__(self, dist, device):
        self.dist = dist
        self.device = device
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def all_sizes(self, n):
        import torch
        t = torch.tensor([int(n)], dtype=torch.int64, device=self.device)
        out = torch.empty(self.world, dtype=torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(out, t)
        return [int(x) for x in out.cpu()]

    def all_gather_var(self, mine, sizes, dtype):
        """mine: 1-D tensor of sizes[rank] elements. Returns a 1-D tensor = concatenation over ranks."""
        import torch
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + int(sz))
        full = torch.empty(max(offs[-1], 1), dtype=dtype, device=self.device)
        if sizes[self.rank]:
            full[offs[self.rank]:offs[self.rank + 1]].copy_(mine)
        works = [self.dist.broadcast(full[offs[r]:offs[r + 1]], src=r, async_op=True) for r in range(self.world) if sizes[r]]
        for w in works:
            if w is not None:
                w.wait()
        return full[:offs[-1]]


def run_stage(engine, lens, exchange=None, iters=CONSISTENCY_ITERS, torch_mod=None):
    """CalcPosteriors + Consistency for this rank. engine: muscle_amd._lib.MpcGpu with set_hmm and
    set_seqs done. exchange: TorchExchange or None (single GPU). Returns this rank's [k0,k1)."""
    n = len(lens)
    npairs = n * (n - 1) // 2
    if exchange is None or exchange.world == 1:
        engine.calc_posteriors(0, npairs)
        engine.build_store()
        if n >= 3:  # mpcflat.cpp:176
            for _ in range(iters):
                engine.cons_iter(0, npairs)
                engine.cons_commit()
        engine.synchronize()
        return 0, npairs
    torch = torch_mod
    cuts = shard_bounds(lens, exchange.world)
    k0, k1 = cuts[exchange.rank], cuts[exchange.rank + 1]
    # ---- stage A on my shard, then all-gather the packed shards
    engine.calc_posteriors(k0, k1)
    nbytes, _ = engine.shard_info()
    sizes = exchange.all_sizes(nbytes)
    mine = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=exchange.device)
    engine.shard_export(mine.data_ptr())
    full = exchange.all_gather_var(mine[:nbytes], sizes, torch.uint8)
    _sync(torch, exchange.device)
    engine.store_import(cuts[:-1], cuts[1:], sizes, full.data_ptr())
    engine._keepalive = full  # dev_all must outlive the store
    # ---- relax on my shard, all-gather the values, commit everywhere
    if n >= 3:
        first, count = engine.values_slice(k0, k1)
        counts = exchange.all_sizes(count)
        for _ in range(iters):
            engine.cons_iter(k0, k1)
            v = torch.empty(max(count, 1), dtype=torch.float32, device=exchange.device)
            engine.values_export(first, count, v.data_ptr())
            allv = exchange.all_gather_var(v[:count], counts, torch.float32)
            _sync(torch, exchange.device)
            engine.values_import(0, int(allv.numel()), allv.data_ptr())
            engine.cons_commit()
    engine.synchronize()
    return k0, k1


def _sync(torch, device):
    if torch is not None and str(device).startswith("cuda"):
        torch.cuda.synchronize()
