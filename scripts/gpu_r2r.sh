#!/bin/bash
# Round 2, GPU call R: the other sizes with the end-of-round build (bench lines, parity digests where fixtures exist, end-to-end runs).
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; LOG=gpurun_out/r2r.log; : > $LOG
line() { python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$1', {k:d.get(k) for k in ('value','ms_per_step','parity_digest')}, {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if v})
except Exception as e: print('$1', l[:400])"; }
for shape in "32 150 20" "256 300 5" "512 400 2" "2000 400 1" "300 1000 2" "100 3000 2"; do
  set -- $shape
  timeout 900 python -u bench.py --n $1 --len $2 --steps $3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -E "^\{" | tail -1 | line "n=$1 L=$2" | tee -a $LOG
done
timeout 300 python -u diag/e2e_named.py super7dm_2000x250_b32 16 2>&1 | tail -1 | tee -a $LOG
MUSCLE_GPU_SHRUB_CONTEXTS=1 timeout 300 python -u diag/e2e_named.py super7dm_2000x250_b32 16 2>&1 | tail -1 | tee -a $LOG
MUSCLE_GPU_DEVICES=0,0 timeout 300 python -u diag/e2e.py 1000 400 16 gpu 2>&1 | tail -1 | tee -a $LOG
