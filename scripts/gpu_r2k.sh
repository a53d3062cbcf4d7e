#!/bin/bash
# Round 2, GPU call K: rocprofv3 kernel stats of one -align 1000x400 run of muscle_gpu (the join kernels of the tail).
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out; LOG=gpurun_out/r2k.log; : > $LOG
python -c "
import sys; sys.path.insert(0,'.')
from muscle_amd.synth import make_family, write_fasta
write_fasta('/tmp/in.fa', make_family(1000,400,seed=1))"
rm -rf gpurun_out/prof_e2e
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e2e -o r -- $R/hostcxx/_build/muscle_gpu -align /tmp/in.fa -output /tmp/out.afa -threads 16 -quiet > /dev/null 2>&1 )
md5sum /tmp/out.afa | tee -a $LOG
f=$(find gpurun_out/prof_e2e -name "*kernel_stats.csv" | head -1)
head -30 $f | cut -c1-200 | tee -a $LOG
python - <<'PY' | tee -a $LOG
import csv, glob, collections
fn = glob.glob("gpurun_out/prof_e2e/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fn)):
    k = r["Kernel_Name"].split("(")[0][:50]
    acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-52s n=%5d total %8.1f ms  median %.3f  p90 %.3f  max %.3f  top100 sum %.1f" % (k, len(v), sum(v), v2[len(v)//2], v2[int(len(v)*0.9)], v2[-1], sum(v2[-100:])))
PY
cp $f gpurun_out/r2k_kernel_stats.csv
find gpurun_out/prof_e2e -name "*kernel_trace.csv" -delete
