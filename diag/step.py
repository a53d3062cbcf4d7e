"""diag/step.py — step-by-step timing of the C-ABI calls (python -u; every step flushed).
usage: python -u diag/step.py N LEN [oracle]"""
import faulthandler
import os
import sys
import time

faulthandler.enable()
faulthandler.dump_traceback_later(int(os.environ.get("DIAG_DUMP_AFTER", "45")), exit=False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
T0 = time.perf_counter()


def say(*a):
    print("[%7.2fs]" % (time.perf_counter() - T0), *a, flush=True)


import numpy as np  # noqa: E402
import _golden as G  # noqa: E402
from muscle_amd._lib import MpcGpu  # noqa: E402
from muscle_amd.synth import make_family  # noqa: E402

n, L = int(sys.argv[1]), int(sys.argv[2])
seqs = make_family(n, L, seed=3)
say("imports done; n=%d L=%d" % (n, L))
g = MpcGpu(0)
say("create ok:", g.version())
s, t, m, i, thr = G.hmm_tables("hmm_amino")
g.set_hmm(s, t, m, i, thr, -1)
say("set_hmm ok")
g.set_seqs(seqs)
say("set_seqs ok, pairs", g.npairs)
g.calc_posteriors()
say("calc_posteriors ok")
ea = g.get_ea().copy()
nnz = g.get_nnz()
say("ea[:4]", ea[:4], "nnz sum", int(nnz.sum()))
g.build_store()
say("build_store ok")
if n >= 3:
    for it in range(2):
        g.cons_iter()
        g.cons_commit()
        g.synchronize()
        say("cons iter", it, "ok")
st = g.get_sparse_range(0, min(g.npairs, 50))
say("get_sparse_range ok")
say("timers", g.timers_get())
if len(sys.argv) > 3:
    import _parity as P
    want = P.run_oracle(seqs, threads=8)
    say("oracle done")
    got = P.run_lib(seqs)
    P.assert_same(got, want, "diag")
    say("PARITY OK vs oracle")
g.close()
say("closed")
faulthandler.cancel_dump_traceback_later()
