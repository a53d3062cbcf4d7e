#!/bin/bash
# A/B of relax_band_kernel geometries / tile shapes on real data (rdrp): one condensed bench line per variant.
# usage: gpurun -- 'bash scripts/sweep_rdrp.sh <tag> <n> "K=V K=V" "K=V" ...'   (each argument = one variant's environment; "-" = defaults)
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; N=$2; shift 2
OUT=$R/gpurun_out; mkdir -p $OUT; LOG=$OUT/$TAG.log; : > $LOG
export PYTHONUNBUFFERED=1
for v in "$@"; do
  echo "== $v" | tee -a $LOG
  ( [ "$v" != "-" ] && export $v; MPCGPU_TRACE=${TRACE:-0} timeout 900 python -u $R/bench.py --fasta $R/tests/golden/rdrp_first1000.fa.gz --n $N --steps 1 --warmup 1 --no-cpu-baseline ${EXTRA:-} 2>&1 ) \
    | tee -a $LOG.full | python $R/scripts/benchline.py | tee -a $LOG
done
echo "== done t=$SECONDS" | tee -a $LOG
