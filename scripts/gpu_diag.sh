#!/bin/bash
# Step-by-step GPU diagnostic with short per-step timeouts (first-light / hang hunting).
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/diag.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1 MPCGPU_TRACE=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $LOG; "$@" 2>&1 | tee -a $LOG | tail -40; echo "=== rc=${PIPESTATUS[0]} (t=$SECONDS)" | tee -a $LOG; }
step cat /sys/fs/cgroup/cpu.max
step timeout 30 diag/hello
step timeout 60 python -u diag/step.py 2 5
step timeout 60 python -u diag/step.py 3 40
step timeout 60 python -u diag/step.py 3 150
step timeout 120 python -u diag/step.py 12 180 oracle
