#!/bin/bash
# Round 2, GPU call G: run-list reduction of the join posteriors (build_post_heads_kernel): GPU suite, end-to-end -align 1000x400 with the library's timers.
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r2g.log; : > $LOG
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $R/$LOG; "$@" 2>&1 | tee -a $R/$LOG | tail -${TAILN:-14}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $R/$LOG; return $rc; }
TAILN=12 step timeout 900 python -u -m pytest tests -m gpu -q
TAILN=40 step env MPCGPU_TRACE_HOST=1 MUSCLE_GPU_TIMING=1 timeout 300 python -u diag/e2e.py 1000 400 16 gpu
TAILN=40 step env MUSCLE_GPU_TIMING=1 timeout 300 python -u diag/e2e.py 1000 400 16 gpu
