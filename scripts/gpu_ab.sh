#!/bin/bash
# A/B: post_kernel LDS sort capacity (occupancy vs global-scratch sorts), stage A only matters
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/ab.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for cap in 1024 256 2; do
  echo "=== MPCGPU_POST_SORT_CAP=$cap" | tee -a $LOG
  MPCGPU_POST_SORT_CAP=$cap MPCGPU_TRACE=0 timeout 150 python -u bench.py --n 400 --len 400 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | tail -1 | tee -a $LOG | cut -c1-120
done
echo "=== default" | tee -a $LOG
timeout 150 python -u bench.py --n 400 --len 400 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | tail -1 | tee -a $LOG | cut -c1-120
