#!/bin/bash
# Round 2, GPU call L: post_rows_kernel EA rows (explicit prefix, pipelined row chain): GPU parity suite, phase clocks, bench line.
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; LOG=gpurun_out/r2l.log; : > $LOG
timeout 900 python -u -m pytest tests -m gpu -q 2>&1 | tail -4 | tee -a $LOG
MPCGPU_POST_PROFILE=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity 2>&1 | grep -E "post_rows_kernel" | head -2 | tee -a $LOG
timeout 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -E "^\{" | tail -1 | tee gpurun_out/bench_r2l.json | cut -c1-1200 | tee -a $LOG
