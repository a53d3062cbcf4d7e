#!/bin/bash
# Round 2, GPU call H: calc_aln_quad_kernel (several waves, rows in registers, staged traceback) — alignment tests, -align 1000x400 with timers.
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
timeout 600 python -u -m pytest tests -m gpu -q -k "aln or msa or dropin or align" 2>&1 | tail -3
MUSCLE_GPU_TIMING=1 timeout 300 python -u diag/e2e.py 1000 400 16 gpu 2>&1 | grep -E "BuildPost|CalcAln|gpu:|library" | tee gpurun_out/r2h.log
