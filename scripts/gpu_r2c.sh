#!/bin/bash
# Round 2, GPU call C: select hazards (pkbench), full GPU suite with the new LOG_ADD, bench, relax time split (DIAG variants).
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r2c.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $LOG; "$@" 2>&1 | tee -a $LOG | tail -${TAILN:-60}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $LOG; return $rc; }
TAILN=50 step timeout 120 diag/pkbench
TAILN=8 step timeout 600 python -u -m pytest tests -m gpu -q -x
for v in "X=default" "MPCGPU_RELAX_DIAG=1" "MPCGPU_RELAX_DIAG=2"; do
  echo "=== bench variant $v (t=$SECONDS)" | tee -a $LOG
  env $v timeout 150 python -u bench.py --n 1000 --len 400 --steps 2 --warmup 1 --no-cpu-baseline --no-parity 2>&1 | grep -E "^\{" | tail -1 | tee -a $LOG | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'])"
done
