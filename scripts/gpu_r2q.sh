#!/bin/bash
# Round 2, GPU call Q: counter passes of the end-of-round kernels at 1000 x L~400 (FETCH_SIZE, WRITE_SIZE, two SQ passes; each in its own
# run, kernel trace only) -> profiles/pmc_traffic.json through scripts/pmc_summary.py.
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r2q.log; : > $LOG
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $R/$LOG; "$@" 2>&1 | tee -a $R/$LOG | tail -${TAILN:-14}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $R/$LOG; return $rc; }
rm -rf $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq $OUT/prof_sq2
B="python -u $R/bench.py --n 1000 --len 400 --steps 1 --warmup 0 --no-cpu-baseline --no-parity"
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/prof_fetch -o r -- $B )
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/prof_write -o r -- $B )
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$OUT/prof_sq -o r -- $B )
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$OUT/prof_sq2 -o r -- $B )
step python scripts/pmc_summary.py 1000 400 $OUT/prof_fetch $OUT/prof_write $OUT/pmc_traffic.json
for d in fetch write sq sq2; do f=$(find $OUT/prof_$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/pmc_${d}_counter_collection.csv; done
ls -la $OUT/*.csv $OUT/pmc_traffic.json | tee -a $LOG
