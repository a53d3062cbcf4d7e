#!/bin/bash
# Round 2, GPU call A: (1) FP32 VALU issue-rate microbenchmark (plain vs packed), (2) the opt-in kernel variants'
# parity test, (3) A/B of the relax / fb variants at 1000 x 400, (4) rocprofv3 kernel stats + FETCH/WRITE + SQ passes
# of the DEFAULT (shipped) kernels.
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r2a.log; : > $LOG
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $R/$LOG; "$@" 2>&1 | tee -a $R/$LOG | tail -${TAILN:-14}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $R/$LOG; return $rc; }
step timeout 60 diag/pkbench
MPCGPU_TEST_OPT_IN=1 step timeout 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -k opt_in_variants
for v in "X=default" "MPCGPU_RELAX_DBUF=1" "MPCGPU_RELAX_PF=1" "MPCGPU_RELAX_DBUF=1 MPCGPU_RELAX_PF=1" "MPCGPU_FB_OCC4=1" "MPCGPU_PAD=rows"; do
  echo "=== bench variant $v (t=$SECONDS)" | tee -a $LOG
  env $v timeout 150 python -u bench.py --n 1000 --len 400 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | tail -1 | tee -a $LOG | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'])"
done
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq $OUT/prof_sq2
B="python -u $R/bench.py --n 1000 --len 400 --steps 1 --warmup 0 --no-cpu-baseline"
( cd /tmp && TAILN=3 step timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats -o r -- python -u $R/bench.py --n 1000 --len 400 --steps 2 --warmup 1 --no-cpu-baseline )
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/prof_fetch -o r -- $B )
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/prof_write -o r -- $B )
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$OUT/prof_sq -o r -- $B )
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$OUT/prof_sq2 -o r -- $B )
find $OUT/prof_stats -name "*kernel_trace.csv" -size +20M -delete
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); do head -14 $f | cut -c1-220 | tee -a $LOG; done
step python scripts/pmc_summary.py 1000 400 $OUT/prof_fetch $OUT/prof_write $OUT/pmc_traffic.json
python - <<'PY' 2>&1 | tee -a $LOG
import csv, glob, collections
for d in ("gpurun_out/prof_sq", "gpurun_out/prof_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k, {a: "%.4g" % b for a, b in sorted(v.items())})
PY
