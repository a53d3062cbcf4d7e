"""diag: host wall time of the FIRST cons_iter of a store (it cuts the band tiles: ~14 ms of small kernels at 1000 x 400) against the
second, step after step — with and without torch in the process. usage: python diag/cut_stall.py [torch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if "torch" in sys.argv[1:]:
    import torch
    torch.zeros(1, device="cuda:0")
import ctypes
def null_stream_op():
    """what torch's first tensor does, without torch: an allocation and a fill on the NULL stream"""
    hip = ctypes.CDLL("libamdhip64.so")
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 21)) == 0
    assert hip.hipMemset(p, 0, ctypes.c_size_t(1 << 21)) == 0
    assert hip.hipDeviceSynchronize() == 0
if "nullbefore" in sys.argv[1:]:
    null_stream_op()
import numpy as np
import _golden as G
from muscle_amd._lib import MpcGpu
from muscle_amd.synth import make_family
seqs = make_family(1000, 400, seed=1)
g = MpcGpu(0)
g.set_hmm(*G.hmm_tables())
g.set_seqs(seqs)
if "nullafter" in sys.argv[1:]:
    null_stream_op()
if "notimers" in sys.argv[1:]:
    g.timers_enable(False)
out = []
for step in range(8):
    g.calc_posteriors(); g.build_store(); g.synchronize()
    t0 = time.perf_counter(); g.cons_iter(); g.synchronize(); t1 = time.perf_counter()
    g.cons_commit(); g.synchronize()
    t2 = time.perf_counter(); g.cons_iter(); g.synchronize(); t3 = time.perf_counter()
    g.cons_commit(); g.synchronize()
    out.append((1e3 * (t1 - t0), 1e3 * (t3 - t2)))
print(" ".join("%.0f/%.0f" % o for o in out[1:]), " (first / second relax iteration incl. the cut, ms)")
