#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/quick5.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $LOG; "$@" 2>&1 | tee -a $LOG | tail -${TAILN:-30}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $LOG; return $rc; }
step timeout 170 python -u diag/e2e.py 1000 400 16 gpu
