"""diag/e2e_super7.py — end-to-end `muscle -super7` (BASELINE config 5 path: guide tree -> shrubs -> MPCFlat per
shrub -> PProg joins) wall time: reference binary vs muscle_gpu, same input and balanced guide tree, plus
identity of the outputs. usage: python diag/e2e_super7.py N LEN [shrub_size] [threads] [gpu]
MUSCLE_BIN_GPU / MUSCLE_BIN_REF override the binaries (e.g. the emulator build for a dry run)."""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from muscle_amd.hostinfo import usable_cores  # noqa: E402
from muscle_amd.synth import make_family, write_fasta  # noqa: E402
import _msa  # noqa: E402

n, L = int(sys.argv[1]), int(sys.argv[2])
shrub = sys.argv[3] if len(sys.argv) > 3 else "32"
th = int(sys.argv[4]) if len(sys.argv) > 4 else usable_cores()
only_gpu = len(sys.argv) > 5 and sys.argv[5] == "gpu"
seqs = make_family(n, L, seed=13)
res = {}
with tempfile.TemporaryDirectory() as d:
    fa = os.path.join(d, "in.fa")
    write_fasta(fa, seqs)
    with open(os.path.join(d, "tree.nwk"), "w") as f:
        f.write(_msa._balanced_newick(0, n) + ";\n")
    for name, binary in (("gpu", os.environ.get("MUSCLE_BIN_GPU", _msa.GPU_MUSCLE)),
                         ("ref", os.environ.get("MUSCLE_BIN_REF", _msa.REF_MUSCLE))):
        if name == "ref" and only_gpu:
            continue
        out = os.path.join(d, name + ".afa")
        t0 = time.perf_counter()
        subprocess.run([binary, "-super7", fa, "-output", out, "-threads", str(th), "-quiet", "-guidetreein", "tree.nwk",
                        "-shrub_size", shrub], check=True, cwd=d, stdout=subprocess.DEVNULL,
                       stderr=None if os.environ.get("MUSCLE_GPU_TIMING") else subprocess.DEVNULL, timeout=3000)
        dt = time.perf_counter() - t0
        res[name] = (dt, hashlib.md5(open(out, "rb").read()).hexdigest())
        print("%s: -super7 %d x L~%d, shrubs of %s, %d threads: %.2f s  md5 %s" % (name, n, L, shrub, th, dt, res[name][1]), flush=True)
if "ref" in res:
    print("identical:", res["gpu"][1] == res["ref"][1], " speedup %.1fx" % (res["ref"][0] / res["gpu"][0]), flush=True)
