#!/bin/bash
# Round 2, GPU call B: instruction-cost table (diag/pkbench), relax geometry trace at 1000 x 400, config-2 digest test.
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r2b.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $LOG; "$@" 2>&1 | tee -a $LOG | tail -${TAILN:-60}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $LOG; return $rc; }
step timeout 120 diag/pkbench
step timeout 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -k "baseline_configs and n256"
MPCGPU_TRACE=1 MPCGPU_RELAX_DBUF=1 timeout 200 python -u bench.py --n 1000 --len 400 --steps 1 --warmup 0 --no-cpu-baseline --no-parity 2>&1 | grep -E "relax tiled|pad|geometry" | sort | uniq -c | tee -a $LOG
