#!/bin/bash
# Round 2, GPU call N: per-join overhead (one staged upload, one result copy, no events unless asked, zeroing inside the generating kernel):
# GPU suite, -super7 10000x250 (config 5), -align 1000x400.
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; LOG=gpurun_out/r2n.log; : > $LOG
timeout 900 python -u -m pytest tests -m gpu -q 2>&1 | tail -4 | tee -a $LOG
timeout 600 python -u diag/e2e_named.py super7dm_10000x250_b32 16 2>&1 | tail -1 | tee -a $LOG
MUSCLE_GPU_TIMING=1 timeout 600 python -u diag/e2e_named.py super7dm_10000x250_b32 16 2>&1 | grep -v "consistency store" | tail -12 | tee -a $LOG
timeout 300 python -u diag/e2e.py 1000 400 16 gpu 2>&1 | tail -1 | tee -a $LOG
timeout 300 python -u diag/e2e_named.py super7dm_2000x250_b32 16 2>&1 | tail -1 | tee -a $LOG
