"""Latency of mpcgpu_align_pairs for tiny lists (what UClust::Search / AlignPairFlat send): wall time per call and the device
time of each kernel family, for 1 and 8 pairs of L x L. Run on the GPU box: python diag/pair_latency.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muscle_amd._lib import MpcGpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import _golden as G  # hmm tables only (no oracle compute)

def main():
    rng = np.random.default_rng(5)
    s, t, m, i, thr = G.hmm_tables()
    for L in (50, 150, 400):
        seqs = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), L)) for _ in range(16)]
        g = MpcGpu(0)
        g.set_hmm(s, t, m, i, thr)
        g.set_seqs_registry(seqs)
        for npairs in (1, 8):
            a = list(range(npairs)); b = [x + 8 for x in a]
            for timers in (False, True):
                g.timers_enable(timers)
                g.timers_reset()
                for _ in range(20): g.align_pairs(a, b)
                g.timers_reset()
                t0 = time.perf_counter()
                K = 300
                for _ in range(K): g.align_pairs(a, b)
                dt = (time.perf_counter() - t0) / K
                line = "L=%d pairs=%d timers=%d: %.3f ms per call" % (L, npairs, timers, dt * 1e3)
                if timers:
                    tm = g.timers_get()
                    line += "  " + ", ".join("%s %.3f" % (k, v[0] / K) for k, v in tm.items() if v[1])
                print(line, flush=True)
        g.close()
main()
