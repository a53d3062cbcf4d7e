"""N>1 host path on CPU: world_size-2 gloo run of muscle_amd.mpcflat.run_stage with a recording
stand-in for the device engine (the real compute only exists on the GPU). Checks that the pair
shards tile [0,pairs), that every rank imports the same packed shards (piece by piece, each at its
place), and that the relaxed values of all ranks reach every rank in position order. The partition is
the library's own (mpcgpu_plan_partition, from the emulator build of the library: host code only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from muscle_amd.mpcflat import TorchExchange, run_stage, shard_bounds, pair_lengths, position_pairs, piece_cuts

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
EMU_LIB = os.path.join(EMU_DIR, "libmpcgpu_emu.so")


def _make_emu():
    import subprocess
    subprocess.check_call(["make", "-C", EMU_DIR], stdout=subprocess.DEVNULL)


def test_shard_bounds_tile_and_balance():
    rng = np.random.default_rng(0)
    lens = rng.integers(50, 500, size=97)
    lx, ly = pair_lengths(lens)
    w = (lx + 1) * (ly + 1)
    for world in (1, 2, 3, 8):
        cuts = shard_bounds(lens, world)
        assert cuts[0] == 0 and cuts[-1] == len(lx) and len(cuts) == world + 1
        assert all(a <= b for a, b in zip(cuts[:-1], cuts[1:]))
        loads = [w[a:b].sum() for a, b in zip(cuts[:-1], cuts[1:])]
        assert max(loads) <= 1.05 * (w.sum() / world) + w.max()
    assert shard_bounds([10, 10], 8)[-1] == 1  # more ranks than pairs: empty shards allowed


@pytest.mark.parametrize("world,n", [(8, 1000), (4, 1000), (2, 64), (3, 100), (5, 64), (6, 64), (7, 200), (16, 400), (8, 9), (8, 3)])
def test_block_partition_covers_balances_and_confines(world, n):
    """mpcgpu_plan_partition: every pair exactly once, rank loads within a few percent, and a rank's pairs touch the sequences
    of its blocks only — half of them at 8 ranks (what its partial store then holds)."""
    from muscle_amd._lib import plan_partition
    _make_emu()
    lens = np.random.default_rng(n + world).integers(30, 500, size=n)
    rects, pos = plan_partition(lens, world, EMU_LIB)
    px, py = position_pairs(n, rects)
    assert len(px) == n * (n - 1) // 2 and pos[0] == 0 and pos[-1] == len(px) and all(a <= b for a, b in zip(pos[:-1], pos[1:]))
    assert np.all(px < py) and len(set(zip(px.tolist(), py.tolist()))) == len(px)
    for xa, xb, ya, yb in rects.tolist():
        assert ya >= xb or (xa == ya and xb == yb)
    if len(rects):  # (too few sequences for groups: the contiguous ranges, any balance)
        w = (lens[px] + 1) * (lens[py] + 1)
        loads = [int(w[a:b].sum()) for a, b in zip(pos[:-1], pos[1:])]
        assert max(loads) <= 1.15 * sum(loads) / world
        need = [len(set(px[a:b].tolist()) | set(py[a:b].tolist())) / n for a, b in zip(pos[:-1], pos[1:])]
        bound = {8: 0.52, 4: 0.77, 2: 1.0}.get(world, 0.5 + 1.0 / world + 0.12)
        assert max(need) <= bound, (need, bound)
    cuts = piece_cuts(lens, px, py, pos, 3)
    for r in range(world):
        assert cuts[r][0] == pos[r] and cuts[r][-1] == pos[r + 1] and all(a <= b for a, b in zip(cuts[r][:-1], cuts[r][1:]))


class FakeEngine:
    """Stands in for muscle_amd._lib.MpcGpu on CPU: 'device pointers' are addresses of CPU tensors."""

    def __init__(self, lens):
        self.lens = lens
        self.n = len(lens)
        self.npairs = self.n * (self.n - 1) // 2
        self.log = []
        self.nnz = np.arange(self.npairs) % 5 + 1  # entries per pair
        self.vbase = np.concatenate([[0], np.cumsum(self.nnz)])

    def plan_partition(self, lens, world):
        from muscle_amd._lib import plan_partition
        return plan_partition(lens, world, EMU_LIB)

    def set_pair_order(self, rects):
        self.rects = np.asarray(rects).copy()

    def calc_posteriors(self, k0, k1):
        self.k0, self.k1 = k0, k1
        # shard blob: one byte per entry of each pair, value = pair index mod 251
        self.blob = torch.tensor(np.repeat(np.arange(k0, k1) % 251, self.nnz[k0:k1]).astype(np.uint8))

    def shard_info(self):
        return int(self.blob.numel()), self.blob.data_ptr()

    def shard_entries(self):
        return int(self.nnz[self.k0:self.k1].sum())

    def shard_export(self, ptr):
        import ctypes
        ctypes.memmove(ptr, self.blob.data_ptr(), self.blob.numel())

    def store_import_part(self, k0s, k1s, sizes, offsets, ptr, own_k0, own_k1):
        import ctypes
        segs = sorted(zip(k0s, k1s, sizes, offsets))
        parts = []
        for a, b, sz, off in segs:  # the shards in position order, wherever they lie in the buffer
            buf = (ctypes.c_ubyte * max(int(sz), 1)).from_address(ptr + int(off))
            parts.append(np.frombuffer(buf, np.uint8)[:int(sz)].copy())
        self.imported = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        self.cuts = ([int(a) for a, _, _, _ in segs], [int(b) for _, b, _, _ in segs], [int(c) for _, _, c, _ in segs])
        self.own = (own_k0, own_k1)
        self.values = np.zeros(int(self.vbase[-1]), np.float32)

    def values_slice(self, k0, k1):
        return int(self.vbase[k0]), int(self.vbase[k1] - self.vbase[k0])

    def values_info(self):
        return self.values.ctypes.data, len(self.values)

    def cons_iter(self, k0, k1):
        self.iter = getattr(self, "iter", 0) + 1
        f, c = self.values_slice(k0, k1)
        self.values[f:f + c] = np.arange(f, f + c) + 1000 * self.iter  # only MY slice is fresh

    def values_export(self, first, count, ptr):
        import ctypes
        ctypes.memmove(ptr, self.values[first:first + count].ctypes.data, 4 * count)

    def values_import(self, first, count, ptr):
        import ctypes
        buf = (ctypes.c_float * count).from_address(ptr)
        self.values[first:first + count] = np.frombuffer(buf, np.float32)

    def cons_commit(self):
        self.log.append(self.values.copy())

    def cons_commit_range(self, first, count):
        # every entry must be committed exactly once per iteration; the log gets the store once all of them are
        if not hasattr(self, "committed") or self.committed is None:
            self.committed = np.zeros(len(self.values), np.int32)
            self.store = np.zeros(len(self.values), np.float32)
        self.committed[first:first + count] += 1
        self.store[first:first + count] = self.values[first:first + count]
        if self.committed.min() >= 1:
            assert self.committed.max() == 1, "an entry committed twice"
            self.log.append(self.store.copy())
            self.committed = None

    def build_store(self):
        pass

    def synchronize(self):
        pass


def _worker(rank, world, port, q, lens=(30, 45, 60, 75, 90, 33, 48)):
    _make_emu()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = list(lens)
    eng = FakeEngine(lens)
    k0, k1 = run_stage(eng, lens, TorchExchange(dist, "cpu"), torch_mod=torch)
    q.put((rank, k0, k1, eng.imported, eng.cuts, [v for v in eng.log]))
    dist.destroy_process_group()


def test_run_stage_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, a0, b0, imp0, cuts0, log0), (r1, a1, b1, imp1, cuts1, log1) = res
    npairs = 21
    assert a0 == 0 and b0 == a1 and b1 == npairs
    assert cuts0[0][0] == 0 and cuts0[1][-1] == npairs and cuts0[0][1:] == cuts0[1][:-1]  # the pieces of both ranks tile the positions
    eng = FakeEngine([30, 45, 60, 75, 90, 33, 48])
    want = np.repeat(np.arange(npairs) % 251, eng.nnz).astype(np.uint8)
    assert np.array_equal(imp0, want) and np.array_equal(imp1, want)
    assert cuts0 == cuts1
    total = int(eng.vbase[-1])
    assert len(log0) == 2 and len(log1) == 2
    for it in (1, 2):
        expect = np.arange(total, dtype=np.float32) + 1000 * it
        assert np.array_equal(log0[it - 1], expect) and np.array_equal(log1[it - 1], expect)


@pytest.mark.parametrize("world,lens", [(4, (30, 45, 60, 75, 90, 33, 48)), (8, (30, 45, 60, 75, 90, 33, 48, 52, 41)),
                                        (8, (20, 25, 30))])  # 3 pairs on 8 ranks: five EMPTY shards
def test_run_stage_world4_and_8_gloo(world, lens):
    """the same exchange at the world sizes of the scaling run (4, 8), including ranks whose pair range is empty: every rank ends
    with every shard, in order, and with every value committed exactly once per iteration"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, lens)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=240) for _ in ps], key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    eng = FakeEngine(list(lens))
    npairs = eng.npairs
    want = np.repeat(np.arange(npairs) % 251, eng.nnz).astype(np.uint8)
    total = int(eng.vbase[-1])
    prev_end = 0
    for rank, k0, k1, imp, cuts, log in res:
        assert k0 == prev_end and k1 >= k0
        prev_end = k1
        assert np.array_equal(imp, want), rank
        assert cuts == res[0][4]
        assert len(log) == 2, (rank, len(log))
        for it in (1, 2):
            assert np.array_equal(log[it - 1], np.arange(total, dtype=np.float32) + 1000 * it), (rank, it)
    assert prev_end == npairs
    if len(lens) == 3:
        assert sum(1 for r in res if r[1] == r[2]) >= 5  # empty shards took part


# ---------------------------------------------------------------------------------------------
# Same world_size-2 gloo run with REAL kernel code on both ranks: the product kernel sources compiled
# against the SIMT emulator (tests/emu, test infrastructure; "device pointers" are host addresses,
# so the CPU tensors of the gloo exchange are valid shard / value buffers). Every rank must end with
# the store the oracle computes for the whole problem, bit for bit: pair-sharded stage A, all-gather
# of the packed shards, store import, sharded relax, all-gather of the values, commit — twice.
def _worker_emu(rank, world, port, q, seqs):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _golden as G
    from muscle_amd._lib import MpcGpu
    s, t, m, i, thr = G.hmm_tables()
    g = MpcGpu(0, EMU_LIB)
    g.set_hmm(s, t, m, i, thr)
    g.set_seqs(seqs)
    lens = [len(x) for x in seqs]
    k0, k1 = run_stage(g, lens, TorchExchange(dist, "cpu"), torch_mod=torch)
    final = g.get_sparse_range()
    q.put((rank, k0, k1, [(o.copy(), v.copy()) for o, v in final], g.get_ea().copy()))
    g.close()
    dist.destroy_process_group()


def test_run_stage_world2_gloo_real_kernels():
    import subprocess
    import _parity as P
    from muscle_amd.synth import make_family
    subprocess.check_call(["make", "-C", EMU_DIR], stdout=subprocess.DEVNULL)
    seqs = make_family(7, 24, seed=17)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_emu, args=(r, 2, port, q, seqs)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=600) for _ in ps], key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want_stages, want_ea = P.run_oracle(seqs)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == 21 and 0 < res[0][2] < 21
    for rank, k0, k1, final, ea in res:
        assert np.array_equal(P.bits(ea), P.bits(want_ea)), "rank %d EA" % rank
        for k, ((o1, v1), (o2, v2)) in enumerate(zip(final, want_stages[2])):
            assert np.array_equal(o1, o2) and np.array_equal(v1, v2), "rank %d pair %d" % (rank, k)


@pytest.mark.parametrize("world,n,pieces", [(3, 12, 2), (2, 9, 3)])
def test_run_stage_gloo_real_kernels_block_partition(world, n, pieces):
    """run_stage over gloo with the REAL kernels (emulator) and enough sequences for the block partition: blocks of the pair
    triangle per rank, stage A in pieces with the exchange of a piece under the next, partial stores (world 3: two of three
    sequence groups per rank). Every rank must end with the oracle's store, read back in InitPairs order."""
    import _parity as P
    from muscle_amd.synth import make_family
    from muscle_amd._lib import plan_partition
    _make_emu()
    seqs = make_family(n, 24, seed=23)
    assert len(plan_partition([len(x) for x in seqs], world, EMU_LIB)[0]) > 0
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    os.environ["MPC_PIECES"] = str(pieces)
    try:
        ps = [ctx.Process(target=_worker_emu, args=(r, world, port, q, seqs)) for r in range(world)]
        for p in ps:
            p.start()
        res = sorted([q.get(timeout=900) for _ in ps], key=lambda r: r[0])
        for p in ps:
            p.join(60)
            assert p.exitcode == 0
    finally:
        del os.environ["MPC_PIECES"]
    want_stages, want_ea = P.run_oracle(seqs)
    assert res[0][1] == 0 and res[-1][2] == n * (n - 1) // 2 and all(a[2] == b[1] for a, b in zip(res[:-1], res[1:]))
    for rank, k0, k1, final, ea in res:
        assert np.array_equal(P.bits(ea), P.bits(want_ea)), "rank %d EA" % rank
        for k, ((o1, v1), (o2, v2)) in enumerate(zip(final, want_stages[2])):
            assert np.array_equal(o1, o2) and np.array_equal(v1, v2), "rank %d pair %d" % (rank, k)


def test_two_rank_thread_exchange_dry_run():
    """tests/_torch_exchange_check.py (the -m gpu check of run_stage's exchange through torch tensors) on the SIMT
    emulator with CPU tensors, so the script itself is known to work before it meets a GPU."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    emu_dir = os.path.join(here, "emu")
    subprocess.check_call(["make", "-C", emu_dir], stdout=subprocess.DEVNULL)
    env = dict(os.environ, PYTHONPATH=os.path.dirname(here) + os.pathsep + here, TEC_DEVICE="cpu",
               TEC_LIB=os.path.join(emu_dir, "libmpcgpu_emu.so"))
    r = subprocess.run([sys.executable, "-u", os.path.join(here, "_torch_exchange_check.py")], env=env, cwd=here,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240, text=True)
    assert r.returncode == 0 and "OK two-rank exchange" in r.stdout, r.stdout[-3000:]


def test_bench_py_world2_dry_run():
    """bench.py itself, launched the way the driver launches it for N=2 (torch.distributed.run, one rank per "GPU"),
    in its tests-only dry-run mode: CPU tensors, gloo, the SIMT-emulator library. Checks the N>1 control flow of the
    file — rendezvous, sharded run_stage, barrier, max-over-ranks, the one JSON line on rank 0 — not performance."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    emu_dir = os.path.join(here, "emu")
    subprocess.check_call(["make", "-C", emu_dir], stdout=subprocess.DEVNULL)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MPCGPU_BENCH_DRYRUN=os.path.join(emu_dir, "libmpcgpu_emu.so"), OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--nseqs", "9", "--seqlen", "30"], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["steps"] == 1 and d["config"]["pairs"] == 36
    assert d["unit"] == "pairs/s" and d["value"] > 0 and d["scaling"] == "strong" and "roofline" in d and "cpu_baseline" not in d
    assert d["config"]["parallelism"] == "pair-shard x2"
    mg = d["multi_gpu"]  # what the first run on real GPUs is read by: ranks the backend connected, the partition, every rank's phases
    assert mg["rccl_ranks"] == 2 and mg["backend"] == "gloo" and len(mg["phase_ms_per_rank"]) == 2 and sum(mg["pairs_per_rank"]) == 36
    assert {"stage_a", "exchange_shards", "import_store", "relax", "exchange_values", "commit"} <= set(mg["phase_ms_max_over_ranks"])
    assert "blocks of the pair triangle" in mg["partition"] and max(mg["sequences_held_per_rank"]) <= 9
