for g in pairs 1 2 4 8 16 32; do
  echo "== ORDER=$g"
  MPCGPU_RELAX_ORDER=$g python -u bench.py --n 1000 --len 400 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python scripts/benchline.py
done
cd /tmp
for g in pairs 1; do
  echo "== counters ORDER=$g"
  rm -rf /tmp/pm_$g
  MPCGPU_RELAX_ORDER=$g timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pm_$g -o r -- python -u $GRAFT_REPO_ROOT/bench.py --n 1000 --len 400 --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/scripts/pmc_kernel_sum.py /tmp/pm_$g relax_band_kernel
done
