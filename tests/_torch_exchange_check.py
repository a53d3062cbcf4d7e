"""Run by tests/test_gpu_parity.py::test_two_rank_exchange_through_torch_tensors in a fresh process (torch must
be imported BEFORE libmpcgpu.so so that both bind the one HIP runtime torch ships, as in bench.py).

Two "ranks" = two threads, each with its own library context on cuda:0, run muscle_amd.mpcflat.run_stage —
the code bench.py runs for N>1 — with the collectives served by an in-process stand-in for torch.distributed
(all_gather_into_tensor on CUDA tensors). What this checks on real hardware, without a second GPU: the packed
shards and relaxed values travel through torch-allocated device memory (pointer interop of the torch allocator
and the library), store_import from that memory, the values exchange and the commit — both ranks must end with
the store a single context computes. RCCL itself is not exercised."""
import os
import sys
import threading

import numpy as np
import torch

import _golden as G
from muscle_amd._lib import MpcGpu
from muscle_amd.mpcflat import TorchExchange, run_stage
from muscle_amd.synth import make_family

WORLD = 2
# dry run without a GPU: TEC_DEVICE=cpu TEC_LIB=tests/emu/libmpcgpu_emu.so (the emulator's "device" pointers are host pointers)
DEVICE = os.environ.get("TEC_DEVICE", "cuda:0")
LIB = os.environ.get("TEC_LIB") or None


class ThreadDist:
    """all_gather_into_tensor / batch_isend_irecv / broadcast / rank / world of torch.distributed for WORLD threads of one process"""

    def __init__(self):
        self.bar = threading.Barrier(WORLD)
        self.slots = [None] * WORLD
        self.tl = threading.local()

    def get_rank(self):
        return self.tl.rank

    def get_world_size(self):
        return WORLD

    def all_gather_into_tensor(self, out, t):
        _sync()
        self.slots[self.tl.rank] = t
        self.bar.wait()
        out.copy_(torch.cat([s.reshape(-1) for s in self.slots]))
        _sync()
        self.bar.wait()


    # the grouped point-to-point form of the exchange (TorchExchange.start_gather_segments): P2POp / isend / irecv / batch_isend_irecv
    isend, irecv = "isend", "irecv"

    def P2POp(self, op, tensor, peer):
        return (op, tensor, peer)

    def batch_isend_irecv(self, ops):
        _sync()
        me = self.tl.rank
        self.slots[me] = {peer: t for (op, t, peer) in ops if op == "isend"}
        self.bar.wait()
        for op, t, peer in ops:
            if op == "irecv":
                t.copy_(self.slots[peer][me])
        _sync()
        self.bar.wait()
        return []

    def broadcast(self, t, src, async_op=False):
        _sync()
        if self.tl.rank == src:
            self.slots[src] = t
        self.bar.wait()
        if self.tl.rank != src:
            t.copy_(self.slots[src])
        _sync()
        self.bar.wait()
        return None


def _sync():
    if DEVICE.startswith("cuda"):
        torch.cuda.synchronize()


def snapshot(g):
    return g.get_sparse_range()


def main():
    if DEVICE.startswith("cuda"):
        torch.cuda.set_device(0)
    seqs = make_family(24, 120, seed=77) if DEVICE.startswith("cuda") else make_family(7, 30, seed=77)
    lens = [len(s) for s in seqs]
    tables = G.hmm_tables()
    whole = MpcGpu(0, LIB)
    whole.set_hmm(*tables)
    whole.set_seqs(seqs)
    run_stage(whole, lens, None)
    want = snapshot(whole)
    want_ea = whole.get_ea().copy()
    whole.close()

    dist = ThreadDist()
    got, errs = [None] * WORLD, []

    def rank_main(r):
        try:
            dist.tl.rank = r
            if DEVICE.startswith("cuda"):
                torch.cuda.set_device(0)
            g = MpcGpu(0, LIB)
            g.set_hmm(*tables)
            g.set_seqs(seqs)
            ex = TorchExchange(dist, DEVICE)
            k0, k1 = run_stage(g, lens, ex, torch_mod=torch)
            got[r] = (snapshot(g), g.get_ea().copy(), (k0, k1))
            g.close()
        except BaseException as e:  # noqa: BLE001
            errs.append((r, repr(e)))
            try:
                dist.bar.abort()
            except Exception:  # noqa: BLE001
                pass

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        print("FAIL", errs)
        return 1
    for r in range(WORLD):
        st, ea, (k0, k1) = got[r]
        assert len(st) == len(want)
        for k, ((o1, v1), (o2, v2)) in enumerate(zip(st, want)):
            assert np.array_equal(o1, o2) and np.array_equal(v1, v2), "rank %d pair %d differs" % (r, k)
        assert np.array_equal(ea.view(np.uint32), want_ea.view(np.uint32)), "rank %d EA" % r
    assert got[0][2][1] == got[1][2][0] and got[0][2][0] == 0 and got[1][2][1] == len(want)
    print("OK two-rank exchange through torch CUDA tensors: %d pairs, shards %s %s" % (len(want), got[0][2], got[1][2]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
