"""diag/dropin_diag.py — why does muscle_gpu's final MSA differ? Runs the drop-in binary on the named
sets under several library modes and prints MD5s next to the live reference's and the golden."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from muscle_amd.hostinfo import pin_openmp_team  # noqa: E402

pin_openmp_team()
import subprocess  # noqa: E402
import tempfile  # noqa: E402

import numpy as np  # noqa: E402
import _msa  # noqa: E402
import _parity as P  # noqa: E402
from muscle_amd.synth import write_fasta  # noqa: E402


def fnv(h, b):
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xffffffffffffffff
    return h


def oracle_digests(seqs):
    stages, ea = P.run_oracle(seqs)
    he = fnv(14695981039346656037, np.asarray(ea, np.float32).tobytes())
    h = 14695981039346656037
    for off, val in stages[-1]:
        h = fnv(h, np.ascontiguousarray(off, np.uint32).tobytes())
        h = fnv(h, np.ascontiguousarray(val, np.uint32).tobytes())
    return he, h


def debug_run(binary, name, threads=8):
    seqs, labels, extra = _msa.input_set(name)
    with tempfile.TemporaryDirectory() as d:
        fa, out = os.path.join(d, "in.fa"), os.path.join(d, "out.afa")
        write_fasta(fa, seqs, labels)
        env = dict(os.environ, MUSCLE_GPU_DEBUG="1")
        r = subprocess.run([binary, "-align", fa, "-output", out, "-threads", str(threads), "-quiet"] + extra,
                           cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
        return [ln for ln in r.stderr.decode().splitlines() if "muscle_gpu" in ln]

sets = sys.argv[1:] or ["n8_L60", "perturb", "bb11005"]
modes = [("default", {}),
         ]
gold = _msa.golden_md5()
for name in sets:
    ref = _msa.run_muscle(_msa.REF_MUSCLE, name, threads=8)[0] if os.path.exists(_msa.REF_MUSCLE) else None
    print("%s: golden=%s ref(live)=%s" % (name, gold.get(name), ref), flush=True)
    if name != "perturb":  # the oracle here uses the default HMM tables
        he, hs = oracle_digests(_msa.input_set(name)[0])
        print("   oracle (input order): fnv(EA) %016x fnv(final store) %016x" % (he, hs), flush=True)
    for rep in range(3):
        for ln in debug_run(_msa.GPU_MUSCLE, name):
            print("   dbg run%d %s" % (rep, ln), flush=True)
    for label, env in modes:
        old = dict(os.environ)
        th = int(env.pop("_threads", "8"))
        os.environ.update(env)
        try:
            for rep in range(2):
                md5 = _msa.run_muscle(_msa.GPU_MUSCLE, name, threads=th)[0]
                print("   %-12s run%d %s %s" % (label, rep, md5, "OK" if md5 == gold.get(name) else "DIFF"), flush=True)
        finally:
            os.environ.clear()
            os.environ.update(old)
