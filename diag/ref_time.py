#!/usr/bin/env python3
"""Times the COMPILED REFERENCE (oracle/_ref/libmuscle_ref.so: MPCFlat::CalcPosteriors + 2 x ConsIter, OpenMP) on the host cores
of the box it runs on, at a size where nothing is extrapolated — and, in the same process order, on the 128-sequence sample
bench.py's cpu_baseline uses, so that the extrapolation formula's error at the timed size is stated, not assumed.
TEST / MEASUREMENT INFRASTRUCTURE: never part of the product path.
  python diag/ref_time.py --n 512 [--len 400] -> one JSON line (commit it under profiles/)."""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(args):
    n, length, seed, cores = args
    import _ref as R
    from muscle_amd.synth import make_family
    seqs = make_family(max(n, 128), length, seed=seed)[:n]
    R.init_hmm(False, 0)
    L = R.lib()
    arr = (C.c_char_p * n)(*[s.encode() for s in seqs])
    assert L.ref_mpc_begin(n, arr, cores) == 0
    t0 = time.perf_counter(); L.ref_mpc_calc_posteriors(); ta = time.perf_counter() - t0
    t0 = time.perf_counter(); L.ref_mpc_cons_iter(0); L.ref_mpc_cons_iter(1); tb = time.perf_counter() - t0
    return {"n": n, "pairs": n * (n - 1) // 2, "stage_a_s": ta, "relax_2it_s": tb}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--len", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from muscle_amd.hostinfo import pin_openmp_team, usable_cores
    pin_openmp_team()
    cores = usable_cores()
    ctx = mp.get_context("spawn")  # ref_mpc_begin: once per process
    out = {"cores": cores, "family": "synthetic L~%d seed %d (the first n sequences of bench.py's family)" % (a.len, a.seed)}
    for key, n in (("sample", 128), ("timed", a.n)):
        with ctx.Pool(1) as pool:
            out[key] = pool.map(run, [(n, a.len, a.seed, cores)])[0]
    s, t = out["sample"], out["timed"]
    per_pair = s["stage_a_s"] / s["pairs"]
    per_triple = s["relax_2it_s"] / (2.0 * s["pairs"] * (s["n"] - 2))
    pred = t["pairs"] * (per_pair + 2.0 * (t["n"] - 2) * per_triple)
    meas = t["stage_a_s"] + t["relax_2it_s"]
    out["timed"]["pairs_per_s"] = t["pairs"] / meas
    out["extrapolated_from_sample_s"] = pred
    out["extrapolation_error"] = pred / meas - 1.0
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu"] = [ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
