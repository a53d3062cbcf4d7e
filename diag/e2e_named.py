"""diag/e2e_named.py — wall time and output MD5 of muscle_gpu (and optionally the reference binary) on a named input set of
tests/_msa.py (e.g. super7dm_10000x250_b32 = BASELINE config 5: -super7 with a precomputed distance matrix).
usage: python diag/e2e_named.py NAME [threads] [ref]      environment (MUSCLE_GPU_*) is passed through"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _msa  # noqa: E402
from muscle_amd.hostinfo import usable_cores  # noqa: E402

name = sys.argv[1]
th = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else usable_cores()
golden = _msa.golden_md5().get(name)
for label, binary in (("gpu", _msa.GPU_MUSCLE),) + ((("ref", _msa.REF_MUSCLE),) if "ref" in sys.argv[2:] else ()):
    t0 = time.perf_counter()
    md5, _ = _msa.run_muscle(binary, name, threads=th, timeout=6000)
    print("%s: %s, %d threads: %.2f s  md5 %s  %s" % (label, name, th, time.perf_counter() - t0, md5,
          "(no reference MD5 committed)" if golden is None else "== reference" if md5 == golden else "DIFFERS from reference %s" % golden), flush=True)
