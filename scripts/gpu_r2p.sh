#!/bin/bash
# Round 2, GPU call P: other shapes on one GPU (one step each, no CPU baseline): more sequences, longer sequences (H = 16 rows per lane),
# sequences beyond 1024 rows (row-block forward/backward kernels).
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; LOG=gpurun_out/r2p.log; : > $LOG
for shape in "2000 400" "300 1000" "100 3000" "64 6000"; do
  set -- $shape
  echo "=== n=$1 L=$2" | tee -a $LOG
  timeout 600 python -u bench.py --n $1 --len $2 --steps 1 --warmup 0 --no-cpu-baseline --no-parity 2>&1 | grep -E "^\{|rror" | tail -1 | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print({k:d[k] for k in ('value','ms_per_step','kernel_ms_per_step')}); print(d['relax_geometry']['layout'])
except Exception as e: print(l[:600])" | tee -a $LOG
done
