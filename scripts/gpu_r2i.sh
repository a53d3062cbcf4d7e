#!/bin/bash
# Round 2, GPU call I: sensitivity of fb_kernel<7> to resident waves per SIMD (LDS pad: 4 -> 3 -> 2 workgroups per CU), process timing of -align.
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; LOG=gpurun_out/r2i.log; : > $LOG
for pad in 0 45 60; do
  echo "=== MPCGPU_FB_LDS_PAD_KB=$pad" | tee -a $LOG
  MPCGPU_FB_LDS_PAD_KB=$pad timeout 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity 2>/dev/null | grep -E "^\{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k in ('value','ms_per_step','kernel_ms','stage_ms')})" | tee -a $LOG
done
MUSCLE_GPU_TIMING=1 timeout 300 python -u diag/e2e.py 1000 400 16 gpu 2>&1 | grep -E "muscle_gpu\]|gpu:" | tee -a $LOG
