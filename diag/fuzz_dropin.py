"""diag/fuzz_dropin.py — seeded random inputs through `muscle_gpu -align` (hostcxx/_build/muscle_gpu: the reference's objects + the device
stage) and the unmodified compiled reference (oracle/_ref/muscle, which travels to the GPU box), same options, same thread count: the final
MSA files must be equal byte for byte. Inputs: diag/fuzz_parity.py's generator (families of any divergence, tiny sequences, one long one
among short ones, low complexity, mixtures, rdrp picks, nucleotides, duplicates, fragments) at sizes the reference finishes in seconds;
options drawn from -consiters 0..3, -refineiters 0..120, -perturb / -perm (the guide tree's variants, mpcflat.cpp:160-212). What this
covers beyond the pinned sets: the drop-in's host side — RefineIter and ProgressiveAlign on position -> column maps, the joins of a tree level
in one batch — on trees and bipartitions nobody chose. A failure prints the case's seed and goes on; exit status 1 if any case differed.
TEST INFRASTRUCTURE (runs the compiled reference).
With `super7`: BASELINE config 5's path instead (super7.cpp:9-137) — families of 12..160 sequences, a balanced guide tree, shrubs of 3..32
sequences on 1..8 worker contexts, MPCFlat per shrub, then the PProg joins (AlignMSAsFlat on the device).
usage: python diag/fuzz_dropin.py [seconds] [first seed] [super7]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "diag"))
ARGV = sys.argv
import _msa  # noqa: E402
from fuzz_parity import make_case  # noqa: E402
from muscle_amd.hostinfo import usable_cores  # noqa: E402
from muscle_amd.synth import write_fasta  # noqa: E402

BUDGET = float(ARGV[1]) if len(ARGV) > 1 else 300.0
SEED0 = int(ARGV[2]) if len(ARGV) > 2 else 1000
SUPER7 = len(ARGV) > 3 and ARGV[3] == "super7"
GPU_MUSCLE = os.environ.get("FUZZ_MUSCLE_GPU", _msa.GPU_MUSCLE)  # (the emulator build of the drop-in, for trying this script without a GPU)


def run(binary, fa, out, threads, extra, env=None, cmd="-align"):
    subprocess.run([binary, cmd, fa, "-output", out, "-threads", str(threads), "-quiet"] + extra, check=True, timeout=900,
                   cwd=os.path.dirname(fa), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=None if env is None else dict(os.environ, **env))
    with open(out, "rb") as f:
        return f.read()


def super7_case(seed, rng, th, failed, opts_seen):
    from muscle_amd.synth import make_family
    n, L = int(rng.integers(12, 161)), int(rng.integers(15, 140))
    shrub = int(rng.integers(3, 33))
    seqs = make_family(n, L, seed=seed, p_sub=float(rng.choice([0.1, 0.3, 0.5])), p_del=float(rng.uniform(0, 0.08)), p_ins=float(rng.uniform(0, 0.08)))
    extra = ["-guidetreein", "tree.nwk", "-shrub_size", str(shrub)]
    env = {"MUSCLE_GPU_SHRUB_CONTEXTS": str(int(rng.integers(1, 9)))}
    if rng.random() < 0.2:
        env["MUSCLE_GPU_DEVICES"] = "0,0"
    opts_seen["shrub contexts " + env["MUSCLE_GPU_SHRUB_CONTEXTS"]] = opts_seen.get("shrub contexts " + env["MUSCLE_GPU_SHRUB_CONTEXTS"], 0) + 1
    what = "super7 n=%d L=%d shrub=%d %s" % (n, L, shrub, env)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "in.fa")
        write_fasta(fa, seqs)
        with open(os.path.join(d, "tree.nwk"), "w") as f:
            f.write(_msa._balanced_newick(0, n) + ";\n")
        try:
            ref = run(_msa.REF_MUSCLE, fa, os.path.join(d, "ref.afa"), th, extra, cmd="-super7")
        except Exception as e:  # noqa: BLE001
            print("skipped seed %d (%s): the reference: %s" % (seed, what, type(e).__name__), flush=True)
            return False
        try:
            got = run(GPU_MUSCLE, fa, os.path.join(d, "gpu.afa"), th, extra, env, cmd="-super7")
            if got != ref:
                raise AssertionError("MSA differs: md5 %s vs the reference's %s" % (hashlib.md5(got).hexdigest(), hashlib.md5(ref).hexdigest()))
        except Exception as e:  # noqa: BLE001 — report and go on
            failed.append((seed, what))
            print("FAILED seed %d: %s: %s" % (seed, what, e), flush=True)
    return True


def main():
    th = int(os.environ.get("FUZZ_THREADS", "0")) or usable_cores()  # -threads of both binaries
    t0 = time.time()
    seed, cases, skipped, failed = SEED0, 0, 0, []
    opts_seen = {}
    while (time.time() - t0 < BUDGET) if BUDGET > 0 else (seed == SEED0):
        what, seqs, hmm, _ = make_case(seed)
        rng = np.random.default_rng(seed + 3)
        seed += 1
        if SUPER7:
            if super7_case(seed - 1, rng, th, failed, opts_seen):
                cases += 1
            else:
                skipped += 1
            continue
        n, L = len(seqs), max(len(s) for s in seqs)
        # the reference's relax is cubic in n with a linear search per cell: keep its run to seconds
        if isinstance(seqs[0], bytes) or n > 40 or L > 1500 or n * n * L > 400000:
            skipped += 1
            continue
        extra = []
        if rng.random() < 0.4:
            extra += ["-consiters", str(int(rng.integers(0, 4)))]
        if rng.random() < 0.5:
            extra += ["-refineiters", str(int(rng.choice([0, 1, 7, 30, 120])))]
        if rng.random() < 0.3:
            extra += ["-perturb", str(int(rng.integers(1, 100))), "-perm", str(rng.choice(["abc", "acb", "bca"]))]
        env = {"MUSCLE_GPU_DEVICES": "0,0"} if rng.random() < 0.15 else None
        for o in extra[0::2] + (["two contexts"] if env else []):
            opts_seen[o] = opts_seen.get(o, 0) + 1
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "in.fa")
            write_fasta(fa, seqs)
            try:
                ref = run(_msa.REF_MUSCLE, fa, os.path.join(d, "ref.afa"), th, extra)
            except Exception as e:  # noqa: BLE001 — an input the reference itself rejects is not a case
                skipped += 1
                print("skipped seed %d (%s %s): the reference: %s" % (seed - 1, what, extra, type(e).__name__), flush=True)
                continue
            try:
                got = run(GPU_MUSCLE, fa, os.path.join(d, "gpu.afa"), th, extra, env)
                if got != ref:
                    raise AssertionError("MSA differs: md5 %s vs the reference's %s" % (hashlib.md5(got).hexdigest(), hashlib.md5(ref).hexdigest()))
            except Exception as e:  # noqa: BLE001 — report and go on
                failed.append((seed - 1, what, extra))
                print("FAILED seed %d: %s %s: %s" % (seed - 1, what, extra, e), flush=True)
        cases += 1
    print("fuzz_dropin: %d cases (%d generated inputs skipped: too large for the reference, or rejected by it), options %s, seeds %d..%d, %d threads, %.0f s: %s" % (
        cases, skipped, opts_seen, SEED0, seed - 1, th, time.time() - t0,
        "every final MSA equal to the reference's, byte for byte" if not failed else "%d FAILED: %s" % (len(failed), failed)), flush=True)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
