"""diag/e2e.py — end-to-end `muscle -align` wall time: reference binary vs muscle_gpu, same input,
plus identity of the outputs. usage: python diag/e2e.py N LEN [threads]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muscle_amd.hostinfo import usable_cores  # noqa: E402
from muscle_amd.synth import make_family, write_fasta  # noqa: E402

n, L = int(sys.argv[1]), int(sys.argv[2])
th = int(sys.argv[3]) if len(sys.argv) > 3 else usable_cores()
only_gpu = len(sys.argv) > 4 and sys.argv[4] == "gpu"
seqs = make_family(n, L, seed=1)
res = {}
with tempfile.TemporaryDirectory() as d:
    fa = os.path.join(d, "in.fa")
    write_fasta(fa, seqs)
    for name, binary in (("gpu", os.path.join(ROOT, "hostcxx", "_build", "muscle_gpu")),
                         ("ref", os.path.join(ROOT, "oracle", "_ref", "muscle"))):
        if name == "ref" and only_gpu:
            continue
        out = os.path.join(d, name + ".afa")
        t0 = time.perf_counter()
        subprocess.run([binary, "-align", fa, "-output", out, "-threads", str(th), "-quiet"], check=True, cwd=d,
                       stdout=subprocess.DEVNULL,
                       stderr=None if os.environ.get("MUSCLE_GPU_TIMING") else subprocess.DEVNULL, timeout=3000)
        dt = time.perf_counter() - t0
        res[name] = (dt, hashlib.md5(open(out, "rb").read()).hexdigest())
        print("%s: %d x L~%d, %d threads: %.2f s  md5 %s" % (name, n, L, th, dt, res[name][1]), flush=True)
if "ref" in res:
    print("identical:", res["gpu"][1] == res["ref"][1], " speedup %.1fx" % (res["ref"][0] / res["gpu"][0]), flush=True)
