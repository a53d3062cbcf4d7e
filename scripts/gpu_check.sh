#!/bin/bash
# One gpurun call: gated first-light check, GPU parity tests, benches, rocprofv3 kernel trace.
# Every step has its own timeout and is logged unbuffered to gpurun_out/check.log.
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/check.log; : > $LOG
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $R/$LOG; "$@" 2>&1 | tee -a $R/$LOG | tail -${TAILN:-25}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $R/$LOG; return $rc; }
MPCGPU_TRACE=1 DIAG_DUMP_AFTER=50 TAILN=12 step timeout 60 python -u diag/step.py 12 180 oracle || exit 11
step timeout 60 python -u -c "import __graft_entry__ as g; g.smoke()"
step timeout 100 python -u bench.py --n 256 --len 300 --steps 2 --warmup 1 --no-cpu-baseline
step timeout 200 python -u bench.py --n 1000 --len 400 --steps 1 --warmup 1 ${BENCH_EXTRA:-}
( cd /tmp && TAILN=4 step timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_1000 -o r1 -- python -u $R/bench.py --n 1000 --len 400 --steps 1 --warmup 1 --no-cpu-baseline )
find $OUT/prof_1000 -name "*kernel_trace.csv" -size +20M -delete
for f in $(find $OUT/prof_1000 -name "*kernel_stats.csv" | head -1); do head -12 $f | cut -c1-200 | tee -a $LOG; done
step timeout 420 python -u -m pytest tests -m gpu -x -q
