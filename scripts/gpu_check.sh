#!/bin/bash
# One gpurun call: GPU parity tests, smoke, a short bench and a rocprofv3 kernel trace.
# Usage (from the repo root, on the GPU box): bash scripts/gpu_check.sh [N] [LEN] [STEPS]
set -u
N=${1:-1000}; LEN=${2:-400}; STEPS=${3:-2}
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocm-smi"; rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep "Model name"
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench small (256x300)"
timeout 600 python bench.py --n 256 --len 300 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | tee $OUT/bench_256x300.json
echo "== bench ${N}x${LEN}"
timeout 1500 python bench.py --n $N --len $LEN --steps $STEPS --warmup 1 2>&1 | tail -2 | tee $OUT/bench_${N}x${LEN}.json
echo "== rocprofv3 kernel trace (256x300)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_256 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --n 256 --len 300 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_256.log 2>&1 )
find $OUT/prof_256 -name "*stats*" | head; for f in $(find $OUT/prof_256 -name "*kernel_stats*csv" | head -1); do head -20 $f; done
