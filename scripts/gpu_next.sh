#!/bin/bash
# Measurements still owed (prepared while the round's GPU budget was exhausted):
#  1. throughput of the PProg join entry point (mpcgpu_align_msas), SURVEY.md 8f row 1
#  2. end-to-end `muscle -super7` (BASELINE config 5 path), reference vs muscle_gpu
#  3. stage A on structure profiles (.mega) and on long sequences (row-block kernel): kernel times via MPCGPU_TRACE
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/next.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $LOG; "$@" 2>&1 | tee -a $LOG | tail -${TAILN:-12}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $LOG; return $rc; }
MPCGPU_TEST_OPT_IN=1 step timeout 200 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -k opt_in_variants   # correctness of the opt-in variants first
# A/B of the relax variants finished without GPU time at the end of round 1 (dense records = default; two LDS buffers, prefetch, OCC4 = opt-in)
for v in "X=default" "MPCGPU_RELAX_DBUF=1" "MPCGPU_PAD=rows" "MPCGPU_RELAX_PF=1" "MPCGPU_FB_OCC4=1"; do
  echo "=== bench variant $v" | tee -a $LOG
  env $v timeout 150 python -u bench.py --n 1000 --len 400 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{" | tail -1 | tee -a $LOG | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'])"
done
step timeout 120 python -u diag/join_bench.py 400 300 2000 3
step timeout 120 python -u diag/join_bench.py 400 600 2000 2
step timeout 600 python -u diag/e2e_super7.py 2000 250 32
step timeout 300 python -u diag/e2e.py 256 300
MUSCLE_GPU_TIMING=1 step timeout 300 python -u diag/e2e.py 1000 400 16 gpu   # where the 15 s go (hostcxx Stopwatch)
