"""Host wall time per library call on a shrub-sized store (25 sequences of L~250: what -super7 runs 412 times): where the milliseconds
of a small MPCFlat::Run go. usage: python diag/small_store_time.py [n] [len] [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _golden as G
from muscle_amd._lib import MpcGpu
from muscle_amd.synth import make_family
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
L = int(sys.argv[2]) if len(sys.argv) > 2 else 250
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
g = MpcGpu(0)
g.set_hmm(*G.hmm_tables())
acc = {}
def T(name, f):
    t0 = time.perf_counter(); r = f(); acc.setdefault(name, []).append(time.perf_counter() - t0); return r
for r in range(reps):
    seqs = make_family(n, L, seed=100 + r)
    T("set_seqs", lambda: g.set_seqs(seqs))
    T("calc_posteriors", lambda: g.calc_posteriors())
    T("build_store", lambda: g.build_store())
    T("cons_iter 1", lambda: g.cons_iter())
    T("commit 1", lambda: g.cons_commit())
    T("cons_iter 2", lambda: g.cons_iter())
    T("commit 2", lambda: g.cons_commit())
    T("synchronize", lambda: g.synchronize())
for k, v in acc.items():
    v = np.array(v[2:]) * 1e3
    print("%-16s mean %.3f ms  min %.3f  max %.3f" % (k, v.mean(), v.min(), v.max()))
print(g.relax_info()[0])
