#!/bin/bash
# the torch-tensor exchange check only (two ranks as threads on one GPU; see tests/_torch_exchange_check.py)
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/exchange.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "=== pytest -m gpu -k two_rank (t=$SECONDS)" | tee -a $LOG
timeout 35 python -u -m pytest tests -m gpu -q -k two_rank_exchange -rA 2>&1 | tee -a $LOG | tail -15
echo "=== rc=${PIPESTATUS[0]} (t=$SECONDS)" | tee -a $LOG
