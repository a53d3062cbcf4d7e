#!/bin/bash
# the structure-profile (.mega) GPU tests only: library vs reference fixtures / oracle, and final-MSA identity of
# muscle_gpu vs the live reference for .mega inputs
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/mega.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "=== pytest -m gpu -k mega (t=$SECONDS)" | tee -a $LOG
timeout 75 python -u -m pytest tests -m gpu -q -k mega -rA 2>&1 | tee -a $LOG | tail -25
echo "=== rc=${PIPESTATUS[0]} (t=$SECONDS)" | tee -a $LOG
