#!/bin/bash
# SQ counter pass over one bench step (separate from --kernel-trace/--stats runs).
set -u
OUT=gpurun_out; mkdir -p $OUT; R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -rf $OUT/prof_sq
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$OUT/prof_sq -o r -- python -u $R/bench.py --n 1000 --len 400 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -3 )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$OUT/prof_sq2 -o r -- python -u $R/bench.py --n 1000 --len 400 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -3 )
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/prof_sq", "gpurun_out/prof_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k, {a: "%.4g" % b for a, b in sorted(v.items())})
PY
