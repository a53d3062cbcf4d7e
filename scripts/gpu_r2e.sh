#!/bin/bash
# Round 2, GPU call E: workgroup-size A/B of relax_var_kernel (4 / 5 / 6 waves per SIMD).
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r2e.log; : > $LOG
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in "MPCGPU_RELAX_WG=2048" "MPCGPU_RELAX_WG=768"; do
  echo "=== bench variant $v (t=$SECONDS)" | tee -a $LOG
  env $v MPCGPU_TRACE=1 timeout 150 python -u bench.py --n 1000 --len 400 --steps 2 --warmup 1 --no-cpu-baseline 2> $OUT/trace.txt | grep -E "^\{" | tail -1 | tee -a $LOG | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'], d.get('parity_digest'))"
  grep -E "store:|relax var|relax tiled" $OUT/trace.txt | sort | uniq -c | tee -a $LOG
done
