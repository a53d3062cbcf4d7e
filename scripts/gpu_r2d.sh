#!/bin/bash
# Round 2, GPU call D: variable-size records + relax_var_kernel (LDS-DMA): parity first, then A/B against the fixed-size layout.
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r2d.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $LOG; "$@" 2>&1 | tee -a $LOG | tail -${TAILN:-60}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $LOG; return $rc; }
TAILN=8 step timeout 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden_sets or baseline_configs or sharded or relax"
for v in "X=default" "MPCGPU_RELAX_WG=512" "MPCGPU_RELAX_NBUF=1" "MPCGPU_RELAX_WG=512 MPCGPU_RELAX_NBUF=2" "MPCGPU_RELAX_DIAG=1" "MPCGPU_RELAX_DIAG=2" "MPCGPU_PAD=dense"; do
  echo "=== bench variant $v (t=$SECONDS)" | tee -a $LOG
  env $v MPCGPU_TRACE=1 timeout 150 python -u bench.py --n 1000 --len 400 --steps 2 --warmup 1 --no-cpu-baseline 2> $OUT/trace.txt | grep -E "^\{" | tail -1 | tee -a $LOG | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'], d.get('parity_digest'))"
  grep -E "store:|relax var|relax tiled" $OUT/trace.txt | sort | uniq -c | tee -a $LOG
done
