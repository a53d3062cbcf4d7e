"""Host-side mirror of the slice of class MPCFlat (mpcflat.h:16-105) that the hot path touches,
for tests and bench.py: InitSeqs/InitPairs -> CalcPosteriors -> Consistency, single GPU or
pair-sharded over torch.distributed ranks (one process per GPU; RCCL all-gather of the packed
sparse posteriors, then of the relaxed values — SURVEY.md §8e). Compute is ONLY the C-ABI library;
the `engine` argument lets the gloo CPU tests substitute a recording stand-in to check the sharding
and exchange logic without a GPU (there is no CPU compute path in the product).

Sharding: the pair triangle is cut into BLOCKS (sequence group i x group j;
include/mpcgpu.h: mpcgpu_plan_partition) and a rank owns whole blocks, so that its pairs touch — and its
store holds the row-indexed matrices of — the sequences of a few groups only (half of the store at 8 ranks).
Every rank enumerates the pairs rank by rank, block by block (mpcgpu_set_pair_order): a rank's pairs are one
contiguous range of POSITIONS, its packed shard and its relaxed values one segment each. Stage A runs in
pieces; the all-gather of piece p travels while piece p + 1 is computed.
"""
import os
import time

import numpy as np

CONSISTENCY_ITERS = 2  # DEFAULT_CONSISTENCY_ITERS_FLAT, mpcflat.h:12
PIECES = 1             # stage-A pieces of a rank of a sharded run (MPC_PIECES overrides). Pieces hide the all-gather of all but the last under the next piece, but a launch of fb_chain_kernel wants >= ~15 pairs per resident wave: at 8 ranks (62 k pairs, 4096 waves) two pieces cost 19 ms of stage A (profiles/r12b), more than the exchange they hide


def pair_lengths(lens):
    """(LX, LY) arrays over pairs in InitPairs order (mpcflat.cpp:139-159)."""
    lens = np.asarray(lens, np.int64)
    n = len(lens)
    ii, jj = np.triu_indices(n, 1)
    return lens[ii], lens[jj]


def shard_bounds(lens, world):
    """Contiguous pair ranges balanced by DP cells sum (LX+1)(LY+1) in InitPairs order: the partition of rounds 1-5, still what
    mpcgpu_plan_partition falls back to (one rank, too few sequences for groups).
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


def position_pairs(n, rects):
    """(x, y) sequence indices of every pair in POSITION order: the enumeration of mpcgpu_set_pair_order (rectangle by rectangle,
    row-major inside; a triangle xa == ya holds the pairs x < y), or InitPairs order when there are no rectangles."""
    rects = np.asarray(rects, np.int64).reshape(-1, 4)
    if len(rects) == 0:
        ii, jj = np.triu_indices(n, 1)
        return ii.astype(np.int64), jj.astype(np.int64)
    xs, ys = [], []
    for xa, xb, ya, yb in rects:
        if ya >= xb:
            x, y = np.meshgrid(np.arange(xa, xb), np.arange(ya, yb), indexing="ij")
            xs.append(x.ravel())
            ys.append(y.ravel())
        else:
            ii, jj = np.triu_indices(xb - xa, 1)
            xs.append(ii + xa)
            ys.append(jj + xa)
    return np.concatenate(xs).astype(np.int64), np.concatenate(ys).astype(np.int64)


def parse_pieces(spec):
    """MPC_PIECES: a count ("2": equal parts) or the parts' shares of the DP cells ("0.85,0.15": a large first piece, whose exchange the
    small second one hides) -> list of shares that sum to 1."""
    if isinstance(spec, (int, np.integer)):
        return [1.0 / max(int(spec), 1)] * max(int(spec), 1)
    if isinstance(spec, (list, tuple)):
        f = [float(x) for x in spec]
    else:
        t = str(spec).strip()
        if "," not in t and "." not in t:
            return parse_pieces(int(t))
        f = [float(x) for x in t.split(",") if x.strip()]
    tot = sum(f)
    return [x / tot for x in f] if f and tot > 0 else [1.0]


def piece_cuts(lens, px, py, rank_pos, pieces):
    """cut points [rank][piece]: every rank's position range in parts by DP cells sum (LX+1)(LY+1): `pieces` equal parts, or parts
    of the given shares (parse_pieces)."""
    shares = parse_pieces(pieces)
    edges = np.concatenate([[0.0], np.cumsum(shares)])
    lens = np.asarray(lens, np.int64)
    cum = np.concatenate([[0], np.cumsum((lens[px] + 1) * (lens[py] + 1))])
    out = []
    for r in range(len(rank_pos) - 1):
        a, b = rank_pos[r], rank_pos[r + 1]
        c = [a]
        for p in range(1, len(shares)):
            target = cum[a] + int((cum[b] - cum[a]) * float(edges[p]))
            c.append(int(min(max(a + np.searchsorted(cum[a:b + 1], target, side="left"), c[-1]), b)))
        c.append(b)
        out.append(c)
    return out


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
        host needs the numbers to place the segments, so this is the one place of a piece where it waits for a collective (a
        few words; the device is not idle meanwhile only if more work is queued — stage A returns per piece, so it is, for the
        tens of microseconds this takes)."""
        import torch
        k = len(mine)
        t = torch.tensor([int(x) for x in mine], dtype=torch.int64, device=self.device)
        out = torch.empty(self.world * k, dtype=torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(out, t)
        host = out.cpu().reshape(self.world, k)
        cols = [[int(host[r, j]) for r in range(self.world)] for j in range(k)]
        return cols[0] if k == 1 else cols

    def buffer(self, key, count, dtype, owner=None, keep=0):
        """a persistent 1-D device buffer of at least `count` elements (grown geometrically, reused by every step: the gathered
        shards are gigabytes and must not be allocated inside the timed loop). The buffers live ON `owner` (the engine whose store
        they feed; default: this exchange) and die with it — keyed by id(engine) here they were never evicted, and a recycled id
        aliased another engine's buffers. keep: elements at the front that a replacement buffer must still hold (the pieces that
        have arrived already)."""
        import torch
        bufs = (self if owner is None else owner).__dict__.setdefault("_xbufs", {})
        b = bufs.get(key)
        if b is None or b.numel() < count or b.dtype != dtype:
            nb = torch.empty(max(int(count + count // 16), 1), dtype=dtype, device=self.device)
            if b is not None and keep and b.dtype == dtype:
                nb[:keep].copy_(b[:keep])
            b = bufs[key] = nb
        return b

    def all_gather_segments(self, full, sizes, offsets=None):
        """full: 1-D tensor that already holds THIS rank's segment at its final place (offset = sum of the sizes before it, or
        offsets[rank]); every other rank's segment is received straight into its own place: exact sizes, queued together and
        waited for once — no padding to the largest shard, no staging copy."""
        works = self.start_gather_segments(full, sizes, offsets)
        return self.finish_gather_segments(full, sizes, works)

    def start_gather_segments(self, full, sizes, offsets=None):
        """queues the transfers of all_gather_segments and returns their handles: the caller has its own device work (the next
        piece of stage A; the commit of its own slice) run under them before it waits. ONE grouped set of point-to-point transfers
        (batch_isend_irecv: on RCCL one ncclGroupStart/End): a rank sends its segment to each peer and receives each peer's
        segment straight into its place — on xGMI's full mesh the 7 sends of a rank leave over 7 links at once, where `world`
        broadcasts on one communicator run one after the other (the same pattern as the one-process group,
        muscle_amd/csrc/mpcgpu_group.cpp). MPC_EXCHANGE=bcast: the `world` broadcasts of rounds 2-4."""
        if offsets is None:
            offsets = [0]
            for sz in sizes[:-1]:
                offsets.append(offsets[-1] + int(sz))
        seg = [full[int(offsets[r]):int(offsets[r]) + int(sizes[r])] for r in range(self.world)]
        if os.environ.get("MPC_EXCHANGE", "p2p") == "bcast":
            return [self.dist.broadcast(seg[r], src=r, async_op=True) for r in range(self.world) if sizes[r]]
        ops = []
        for d in range(1, self.world):  # peer at distance d: receive from the rank d behind, send to the rank d ahead
            src, dst = (self.rank - d) % self.world, (self.rank + d) % self.world
            if sizes[src]:
                ops.append(self.dist.P2POp(self.dist.irecv, seg[src], src))
            if sizes[self.rank]:
                ops.append(self.dist.P2POp(self.dist.isend, seg[self.rank], dst))
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


def plan(engine, lens, world):
    """(rects, rank_pos, px, py) of the block partition for `world` ranks, cached on the engine (a bench repeats the same step)."""
    key = (world, tuple(int(x) for x in lens))
    c = getattr(engine, "_plan_cache", None)
    if c is None or c[0] != key:
        rects, pos = engine.plan_partition(lens, world)
        px, py = position_pairs(len(lens), rects)
        c = engine._plan_cache = (key, rects, pos, px, py)
    return c[1:]


def run_stage(engine, lens, exchange=None, iters=CONSISTENCY_ITERS, torch_mod=None, pieces=None):
    """CalcPosteriors + Consistency for this rank. engine: muscle_amd._lib.MpcGpu with set_hmm and
    set_seqs done. exchange: TorchExchange or None (single GPU). Returns this rank's range of POSITIONS [k0,k1) in the pair
    order of the run (InitPairs numbers on one GPU). Host seconds per phase go to engine._phase (bench.py reports them)."""
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
    world, rank = exchange.world, exchange.rank
    ph = engine.__dict__.setdefault("_phase", {})

    def lap(name, t0):
        ph[name] = ph.get(name, 0.0) + time.perf_counter() - t0

    # ---- the partition: blocks of the pair triangle; from here on "k" is a position in the run's pair order
    rects, pos, px, py = plan(engine, lens, world)
    engine.set_pair_order(rects)
    k0, k1 = pos[rank], pos[rank + 1]
    spec = pieces if pieces is not None else os.environ.get("MPC_PIECES", PIECES)
    P = len(parse_pieces(spec))
    cuts = piece_cuts(lens, px, py, pos, spec)
    # ---- stage A on my range, piece by piece; the all-gather of piece p (every rank's piece, each straight into its place in ONE
    # persistent gather buffer) travels while piece p + 1 is computed
    seg_k0, seg_k1, seg_bytes, seg_off = [], [], [], []
    counts = [0] * world
    total, pending, full = 0, [], None
    for p in range(P):
        t0 = time.perf_counter()
        engine.calc_posteriors(cuts[rank][p], cuts[rank][p + 1])
        nbytes, _ = engine.shard_info()
        lap("stage_a", t0)
        t0 = time.perf_counter()
        sizes, ents = exchange.all_sizes(nbytes, engine.shard_entries())
        padded = [(s + 15) & ~15 for s in sizes]
        offs = [total + int(sum(padded[:r])) for r in range(world)]
        want = total + int(sum(padded))
        if full is None or full.numel() < want:
            for works in pending:  # (a replacement buffer: what is in flight lands in the old one first)
                exchange.finish_gather_segments(full, [], works)
            pending = []
            guess = total + int(sum(padded)) * (P - p) * 9 // 8 + 4096  # by this piece's size, room for the ones to come
            full = exchange.buffer("shards", guess, torch.uint8, owner=engine, keep=total)
        engine.shard_export(full.data_ptr() + offs[rank])
        pending.append(exchange.start_gather_segments(full, sizes, offs))
        for r in range(world):
            seg_k0.append(cuts[r][p]); seg_k1.append(cuts[r][p + 1]); seg_bytes.append(sizes[r]); seg_off.append(offs[r])
            counts[r] += ents[r]
        total = want
        lap("exchange_shards", t0)
    t0 = time.perf_counter()
    for works in pending:
        exchange.finish_gather_segments(full, [], works)
    _sync(torch, exchange.device)
    lap("exchange_shards", t0)
    t0 = time.perf_counter()
    engine.store_import_part(seg_k0, seg_k1, seg_bytes, seg_off, full.data_ptr(), k0, k1)
    engine._keepalive = full  # dev_all must outlive the store
    lap("import_store", t0)
    # ---- relax on my range, all-gather the values (each rank's slice straight into its place), commit everywhere
    if n >= 3:
        first, count = engine.values_slice(k0, k1)
        assert count == counts[rank], (count, counts)
        # the all-gather of the values runs IN PLACE in the library's values array (position order = rank order: every rank's slice is one
        # segment): the array is wrapped as a tensor of the exchange's device (zero-copy; round 6 — rounds 2-5 copied the own slice out
        # and the peers' slices in, 1.5 GB per iteration)
        vptr, total_v = engine.values_info()
        allv = _wrap(torch, vptr, total_v, exchange.device)
        assert total_v == int(sum(counts)), (total_v, counts)
        for _ in range(iters):
            t0 = time.perf_counter()
            engine.cons_iter(k0, k1)
            engine.synchronize()  # (the own slice is complete before it is sent)
            lap("relax", t0)
            t0 = time.perf_counter()
            works = exchange.start_gather_segments(allv, counts)
            # my own slice is committed (library stream) while the peers' slices arrive (collective stream); theirs follow — every
            # entry once, which equals one commit of everything
            engine.cons_commit_range(first, count)
            exchange.finish_gather_segments(allv, counts, works)
            _sync(torch, exchange.device)
            lap("exchange_values", t0)
            t0 = time.perf_counter()
            if first:
                engine.cons_commit_range(0, first)
            if first + count < total_v:
                engine.cons_commit_range(first + count, total_v - first - count)
            engine.synchronize()
            lap("commit", t0)
    engine.synchronize()
    engine._exchange_seconds = ph.get("exchange_shards", 0.0) + ph.get("exchange_values", 0.0)
    return k0, k1


def _wrap(torch, ptr, count, device):
    """a float32 tensor over `count` floats at address `ptr` of the library's memory, without a copy: __cuda_array_interface__ on a GPU
    (torch.as_tensor aliases it: diag/cai_probe.py), a ctypes buffer on the CPU (the emulator's "device" memory is host memory)."""
    if count == 0:
        return torch.empty(0, dtype=torch.float32, device=device)
    if str(device).startswith("cuda"):
        class _Mem:
            pass
        m = _Mem()
        m.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}
        return torch.as_tensor(m, device=device)
    import ctypes
    return torch.frombuffer((ctypes.c_float * int(count)).from_address(int(ptr)), dtype=torch.float32)


def _sync(torch, device):
    if torch is not None and str(device).startswith("cuda"):
        torch.cuda.synchronize()
