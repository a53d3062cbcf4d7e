#!/bin/bash
# Round 2, GPU call O (final state of the round): smoke, full GPU suite, default bench line with CPU baseline, rocprofv3 kernel stats of the
# bench command, rdrp real-data line, end-to-end runs.
set -u
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/r2o.log; : > $LOG
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp PYTHONUNBUFFERED=1
step() { echo "=== $* (t=$SECONDS)" | tee -a $R/$LOG; "$@" 2>&1 | tee -a $R/$LOG | tail -${TAILN:-14}; rc=${PIPESTATUS[0]}; echo "=== rc=$rc (t=$SECONDS)" | tee -a $R/$LOG; return $rc; }
TAILN=4 step timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
TAILN=6 step timeout 900 python -u -m pytest tests -m gpu -q
echo "=== default bench (t=$SECONDS)" | tee -a $LOG
timeout 400 python -u bench.py 2>/dev/null | grep -E "^\{" | tail -1 | tee $OUT/bench_default.json | cut -c1-2600 | tee -a $LOG
rm -rf $OUT/prof_stats
( cd /tmp && TAILN=3 step timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats -o r -- python -u $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity )
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats_final.csv; head -12 $f | cut -c1-200 | tee -a $LOG; done
find $OUT/prof_stats -name "*kernel_trace.csv" -delete
echo "=== rdrp 1000 (t=$SECONDS)" | tee -a $LOG
timeout 600 python -u bench.py --fasta tests/golden/rdrp_first1000.fa.gz --n 1000 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | grep -E "^\{" | tail -1 | tee $OUT/bench_rdrp1000.json | cut -c1-900 | tee -a $LOG
echo "=== end to end (t=$SECONDS)" | tee -a $LOG
timeout 300 python -u diag/e2e.py 1000 400 16 gpu 2>&1 | tail -1 | tee -a $LOG
timeout 300 python -u diag/e2e.py 256 300 16 gpu 2>&1 | tail -1 | tee -a $LOG
timeout 300 python -u diag/e2e_named.py super7dm_10000x250_b32 16 2>&1 | tail -1 | tee -a $LOG
