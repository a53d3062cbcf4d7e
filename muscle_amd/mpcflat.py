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
    ROCm; gloo in the CPU tests). Shards differ in size, so the all-gather is one grouped set of
    exact-size point-to-point transfers, each segment straight into its final place in ONE
    preallocated buffer (queued together, waited for once): no padding to the largest shard and no
    concatenation copy."""

    def __init__(self, dist, device):
        self.dist = dist
        self.device = device
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def all_sizes(self, *mine):
        """ONE all-gather of this rank's counts (e.g. shard bytes and value count) -> one list per count, indexed by rank. The
        host needs the numbers to size the gather buffers, so this is the one place of a step where it waits for a collective."""
        import torch
        k = len(mine)
        t = torch.tensor([int(x) for x in mine], dtype=torch.int64, device=self.device)
        out = torch.empty(self.world * k, dtype=torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(out, t)
        host = out.cpu().reshape(self.world, k)
        cols = [[int(host[r, j]) for r in range(self.world)] for j in range(k)]
        return cols[0] if k == 1 else cols

    def buffer(self, key, count, dtype, owner=None):
        """a persistent 1-D device buffer of at least `count` elements (grown geometrically, reused by every step: the gathered
        shards are gigabytes and must not be allocated inside the timed loop). The buffers live ON `owner` (the engine whose store
        they feed; default: this exchange) and die with it — keyed by id(engine) here they were never evicted, and a recycled id
        aliased another engine's buffers."""
        import torch
        bufs = (self if owner is None else owner).__dict__.setdefault("_xbufs", {})
        b = bufs.get(key)
        if b is None or b.numel() < count or b.dtype != dtype:
            b = torch.empty(max(int(count + count // 16), 1), dtype=dtype, device=self.device)
            bufs[key] = b
        return b

    def all_gather_segments(self, full, sizes):
        """full: 1-D tensor that already holds THIS rank's segment at its final place (offset = sum of the sizes before it);
        every other rank's segment is received straight into its own place: exact sizes, queued together and waited for
        once — no padding to the largest shard, no staging copy."""
        works = self.start_gather_segments(full, sizes)
        return self.finish_gather_segments(full, sizes, works)

    def start_gather_segments(self, full, sizes):
        """queues the transfers of all_gather_segments and returns their handles: the caller has its own device work (the commit
        of its own slice) run under them before it waits. ONE grouped set of point-to-point transfers (batch_isend_irecv: on RCCL
        one ncclGroupStart/End): a rank sends its segment to each peer and receives each peer's segment straight into its place —
        on xGMI's full mesh the 7 sends of a rank leave over 7 links at once, where `world` broadcasts on one communicator run one
        after the other (the same pattern as the one-process group, muscle_amd/csrc/mpcgpu_group.cpp). MPC_EXCHANGE=bcast: the
        `world` broadcasts of rounds 2-4."""
        import os
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + int(sz))
        if os.environ.get("MPC_EXCHANGE", "p2p") == "bcast":
            return [self.dist.broadcast(full[offs[r]:offs[r + 1]], src=r, async_op=True) for r in range(self.world) if sizes[r]]
        mine = full[offs[self.rank]:offs[self.rank + 1]]
        ops = []
        for d in range(1, self.world):  # peer at distance d: receive from the rank d behind, send to the rank d ahead
            src, dst = (self.rank - d) % self.world, (self.rank + d) % self.world
            if sizes[src]:
                ops.append(self.dist.P2POp(self.dist.irecv, full[offs[src]:offs[src + 1]], src))
            if sizes[self.rank]:
                ops.append(self.dist.P2POp(self.dist.isend, mine, dst))
        return list(self.dist.batch_isend_irecv(ops)) if ops else []

    def finish_gather_segments(self, full, sizes, works):
        for w in works:
            if w is not None:
                w.wait()
        return full[:int(sum(int(sz) for sz in sizes))]

    def all_gather_var(self, mine, sizes, dtype):
        """mine: 1-D tensor of sizes[rank] elements. Returns a 1-D tensor = concatenation over ranks."""
        import torch
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + int(sz))
        full = torch.empty(max(offs[-1], 1), dtype=dtype, device=self.device)
        if sizes[self.rank]:
            full[offs[self.rank]:offs[self.rank + 1]].copy_(mine)
        return self.all_gather_segments(full, sizes)


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
    import time
    torch = torch_mod
    cuts = shard_bounds(lens, exchange.world)
    k0, k1 = cuts[exchange.rank], cuts[exchange.rank + 1]
    t_x = 0.0  # host seconds inside the two exchanges (incl. the waits for them): reported by bench.py as exchange_ms
    # ---- stage A on my shard, then all-gather the packed shards: my shard is written once, at its final place in the
    # persistent gather buffer, and the peers' shards arrive at theirs
    engine.calc_posteriors(k0, k1)
    nbytes, _ = engine.shard_info()
    t0 = time.perf_counter()
    # shard bytes and value counts of every rank in one exchange (the stored cells of a shard are known once its stage A is done)
    sizes, counts = exchange.all_sizes(nbytes, engine.shard_entries())
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    # (the buffers belong to this engine's store and are kept on the engine: one set per engine, freed with it)
    full = exchange.buffer("shards", int(offs[-1]), torch.uint8, owner=engine)
    engine.shard_export(full.data_ptr() + int(offs[exchange.rank]))
    full = exchange.all_gather_segments(full, sizes)
    _sync(torch, exchange.device)
    t_x += time.perf_counter() - t0
    engine.store_import(cuts[:-1], cuts[1:], sizes, full.data_ptr())
    engine._keepalive = full  # dev_all must outlive the store
    # ---- relax on my shard, all-gather the values (each rank's slice straight into its place), commit everywhere
    if n >= 3:
        first, count = engine.values_slice(k0, k1)
        assert count == counts[exchange.rank], (count, counts)
        voffs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        allv = exchange.buffer("values", int(voffs[-1]), torch.float32, owner=engine)
        for _ in range(iters):
            engine.cons_iter(k0, k1)
            t0 = time.perf_counter()
            engine.values_export(first, count, allv.data_ptr() + 4 * int(voffs[exchange.rank]))
            works = exchange.start_gather_segments(allv, counts)
            # my own slice is already in the store's values array: it is committed (library stream) while the peers' slices
            # arrive (collective stream); theirs follow — every entry once, which equals one commit of everything
            engine.cons_commit_range(first, count)
            got = exchange.finish_gather_segments(allv, counts, works)
            _sync(torch, exchange.device)
            t_x += time.perf_counter() - t0
            total = int(got.numel())
            if first:
                engine.values_import(0, first, got.data_ptr())
                engine.cons_commit_range(0, first)
            if first + count < total:
                engine.values_import(first + count, total - first - count, got.data_ptr() + 4 * (first + count))
                engine.cons_commit_range(first + count, total - first - count)
    engine.synchronize()
    engine._exchange_seconds = getattr(engine, "_exchange_seconds", 0.0) + t_x
    return k0, k1


def _sync(torch, device):
    if torch is not None and str(device).startswith("cuda"):
        torch.cuda.synchronize()
