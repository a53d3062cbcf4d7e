#!/bin/bash
# Round 2, GPU call F: full GPU suite (groups of contexts, drop-in with MUSCLE_GPU_DEVICES), real data (rdrp), profiles of the default kernels.
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r2f.log; : > $LOG
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $R/$LOG; "$@" 2>&1 | tee -a $R/$LOG | tail -${TAILN:-14}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $R/$LOG; return $rc; }
TAILN=12 step timeout 900 python -u -m pytest tests -m gpu -q
echo "=== rdrp 1000 (t=$SECONDS)" | tee -a $LOG
MPCGPU_TRACE=1 timeout 600 python -u bench.py --fasta tests/golden/rdrp_first1000.fa.gz --n 1000 --steps 1 --warmup 0 2> $OUT/trace_rdrp.txt | grep -E "^\{" | tail -1 | tee $OUT/bench_rdrp1000.json | cut -c1-1500 | tee -a $LOG
grep -E "store:|relax var|relax tiled" $OUT/trace_rdrp.txt | sort | uniq -c | tee -a $LOG
echo "=== synthetic default with CPU baseline (t=$SECONDS)" | tee -a $LOG
timeout 300 python -u bench.py --steps 3 --warmup 1 2>/dev/null | grep -E "^\{" | tail -1 | tee $OUT/bench_default.json | cut -c1-2500 | tee -a $LOG
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq $OUT/prof_sq2
B="python -u $R/bench.py --n 1000 --len 400 --steps 1 --warmup 0 --no-cpu-baseline --no-parity"
( cd /tmp && TAILN=3 step timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats -o r -- python -u $R/bench.py --n 1000 --len 400 --steps 2 --warmup 1 --no-cpu-baseline --no-parity )
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
        if "relax" in k or "fb_kernel" in k:
            print(k, {a: "%.4g" % b for a, b in sorted(v.items())})
PY
