#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/quick.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $LOG; "$@" 2>&1 | tee -a $LOG | tail -${TAILN:-30}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $LOG; return $rc; }
TAILN=8 step timeout 300 python -u -m pytest tests -m gpu -q -x
step timeout 150 python -u bench.py --n 1000 --len 400 --steps 1 --warmup 1 --no-cpu-baseline
