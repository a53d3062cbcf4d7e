#!/bin/bash
# One gpurun call: gated first-light checks, smoke, benches, rocprofv3 kernel trace, GPU parity tests.
# Every step has its own timeout and is logged unbuffered to gpurun_out/check.log.
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/check.log; : > $LOG
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $LOG; "$@" 2>&1 | tee -a $LOG | tail -25; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $LOG; return $rc; }
MPCGPU_TRACE=1 step timeout 40 python -u diag/step.py 2 5 || exit 10
MPCGPU_TRACE=1 DIAG_DUMP_AFTER=80 step timeout 90 python -u diag/step.py 12 180 oracle || exit 11
step timeout 90 python -u -c "import __graft_entry__ as g; g.smoke()"
step timeout 150 python -u bench.py --n 256 --len 300 --steps 2 --warmup 1 --no-cpu-baseline
tail -1 $LOG > /dev/null
step timeout 320 python -u bench.py --n 1000 --len 400 --steps 1 --warmup 1
( cd /tmp && step timeout 260 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_1000 -o r1 -- python -u $R/bench.py --n 1000 --len 400 --steps 1 --warmup 1 --no-cpu-baseline ) 2>&1 | tail -5
find $OUT/prof_1000 -name "*stats*" | head; for f in $(find $OUT/prof_1000 -name "*kernel_stats*csv" | head -1); do head -20 $f | tee -a $LOG; done
find $OUT/prof_1000 -name "*kernel_trace.csv" -size +20M -delete
step timeout 420 python -u -m pytest tests -m gpu -x -q
