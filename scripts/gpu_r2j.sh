#!/bin/bash
# Round 2, GPU call J: run queue of build_post_reduce_kernel, resident waves per SIMD 1/2/3/4: -align 1000x400 with timers.
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; LOG=gpurun_out/r2j.log; : > $LOG
for w in 2 1 3 4; do
  echo "=== MPCGPU_BP_WAVES=$w" | tee -a $LOG
  MPCGPU_BP_WAVES=$w MUSCLE_GPU_TIMING=1 timeout 300 python -u diag/e2e.py 1000 400 16 gpu 2>&1 | grep -E "reduce|AlignAlns: library|gpu:" | tee -a $LOG
done
