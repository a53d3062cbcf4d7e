#!/bin/bash
# One gpurun call: GPU tests, smoke, bench (with CPU baseline), rocprofv3 kernel stats and PMC passes.
# Every step has its own timeout and is logged unbuffered to gpurun_out/check.log.
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/check.log; : > $LOG
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
N=${BENCH_N:-1000}; L=${BENCH_L:-400}
step() { echo "=== $* (t=$SECONDS)" | tee -a $R/$LOG; "$@" 2>&1 | tee -a $R/$LOG | tail -${TAILN:-25}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $R/$LOG; return $rc; }
TAILN=12 step timeout 400 python -u -m pytest tests -m gpu -q
step timeout 60 python -u -c "import __graft_entry__ as g; g.smoke()"
step timeout 240 python -u bench.py --n $N --len $L --steps 1 --warmup 1
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write
( cd /tmp && TAILN=3 step timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats -o r -- python -u $R/bench.py --n $N --len $L --steps 1 --warmup 1 --no-cpu-baseline )
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/prof_fetch -o r -- python -u $R/bench.py --n $N --len $L --steps 1 --warmup 0 --no-cpu-baseline )
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/prof_write -o r -- python -u $R/bench.py --n $N --len $L --steps 1 --warmup 0 --no-cpu-baseline )
find $OUT/prof_stats -name "*kernel_trace.csv" -size +20M -delete
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); do head -12 $f | cut -c1-200 | tee -a $LOG; done
step python scripts/pmc_summary.py $N $L $OUT/prof_fetch $OUT/prof_write $OUT/pmc_traffic.json
