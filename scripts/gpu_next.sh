#!/bin/bash
# Measurements still owed (prepared while the round's GPU budget was exhausted):
#  1. throughput of the PProg join entry point (mpcgpu_align_msas), SURVEY.md 8f row 1
#  2. end-to-end `muscle -super7` (BASELINE config 5 path), reference vs muscle_gpu
#  3. stage A on structure profiles (.mega) and on long sequences (row-block kernel): kernel times via MPCGPU_TRACE
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/next.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $LOG; "$@" 2>&1 | tee -a $LOG | tail -${TAILN:-12}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $LOG; return $rc; }
step timeout 120 python -u diag/join_bench.py 400 300 2000 3
step timeout 120 python -u diag/join_bench.py 400 600 2000 2
step timeout 600 python -u diag/e2e_super7.py 2000 250 32
step timeout 300 python -u diag/e2e.py 256 300
MUSCLE_GPU_TIMING=1 step timeout 300 python -u diag/e2e.py 1000 400 16 gpu   # where the 15 s go (hostcxx Stopwatch)
