#!/bin/bash
# A/B of relax_band_kernel's cell order (MPCGPU_RELAX_ORDER) on the GPU box: one bench line per value, then one --pmc pass for two of them
# usage: gpurun -- 'bash scripts/sweep_order.sh [values...]'
R=${GRAFT_REPO_ROOT:-$PWD}
VALS=${@:-pairs 1 2 4 8 16}
for g in $VALS; do
  echo "== ORDER=$g"
  MPCGPU_RELAX_ORDER=$g python -u $R/bench.py --n 1000 --len 400 --steps 2 --warmup 1 --no-cpu-baseline --no-real-data 2>/dev/null | python $R/scripts/benchline.py | head -1
done
cd /tmp
for g in pairs 8; do
  echo "== counters ORDER=$g"
  rm -rf /tmp/pm_$g
  MPCGPU_RELAX_ORDER=$g timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pm_$g -o r -- python -u $R/bench.py --n 1000 --len 400 --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-real-data > /dev/null 2>&1
  python $R/scripts/pmc_kernel_sum.py /tmp/pm_$g relax_band_kernel
done
