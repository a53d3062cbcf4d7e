"""diag/dropin_diag.py — why does muscle_gpu's final MSA differ? Runs the drop-in binary on the named
sets under several library modes and prints MD5s next to the live reference's and the golden."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from muscle_amd.hostinfo import pin_openmp_team  # noqa: E402

pin_openmp_team()
import _msa  # noqa: E402

sets = sys.argv[1:] or ["n8_L60", "perturb", "bb11005"]
modes = [("default", {}), ("trace(sync)", {"MPCGPU_TRACE": "1"}), ("gather", {"MPCGPU_RELAX": "gather"}),
         ("threads1", {"_threads": "1"})]
gold = _msa.golden_md5()
for name in sets:
    ref = _msa.run_muscle(_msa.REF_MUSCLE, name, threads=8)[0] if os.path.exists(_msa.REF_MUSCLE) else None
    print("%s: golden=%s ref(live)=%s" % (name, gold.get(name), ref), flush=True)
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
