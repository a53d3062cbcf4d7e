#!/bin/bash
# A/B: bench under two relax workgroup geometries
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/ab.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for wg in 512 1024; do
  echo "=== MPCGPU_RELAX_WG=$wg" | tee -a $LOG
  MPCGPU_RELAX_WG=$wg MPCGPU_TRACE=${TRACE:-0} timeout 150 python -u bench.py --n 1000 --len 400 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "relax tiled|^\{" | tail -3 | tee -a $LOG | cut -c1-200
done
