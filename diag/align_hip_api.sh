#!/bin/bash
# diag/align_hip_api.sh — host-side HIP API time of one `muscle_gpu -align` run (1000 x L~400): rocprofv3 --hip-runtime-trace --stats,
# which API calls the cold run spends its time in (allocations, frees, synchronisations) and the calls longer than 5 ms in order.
# usage: bash diag/align_hip_api.sh <outdir under the repo>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/${1:?outdir}; mkdir -p $OUT
python - <<PY
import sys; sys.path.insert(0, "$R")
from muscle_amd.synth import make_family, write_fasta
write_fasta("/tmp/in1000.fa", make_family(1000, 400, seed=1))
PY
cd /tmp && export TMPDIR=/tmp
time $R/hostcxx/_build/muscle_gpu -align /tmp/in1000.fa -output /tmp/o0.afa -threads 16 -quiet
time $R/hostcxx/_build/muscle_gpu -align /tmp/in1000.fa -output /tmp/o0.afa -threads 16 -quiet
rocprofv3 --hip-runtime-trace --stats --output-format csv -d $OUT -o r -- $R/hostcxx/_build/muscle_gpu -align /tmp/in1000.fa -output /tmp/o1.afa -threads 16 -quiet 2>&1 | grep -v "^[WE]2026"
md5sum /tmp/o0.afa /tmp/o1.afa
f=$(find $OUT -name "*hip_api_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-200
python - <<PY
import csv, glob
a = glob.glob("$OUT/**/*hip_api_trace.csv", recursive=True)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(a[0]))]
rows.sort(); t0 = rows[0][0]
for s, e, f in rows:
    if e - s > 5_000_000: print("%9.3f ms %-32s %9.3f ms" % ((s - t0) / 1e6, f, (e - s) / 1e6))
PY
find $OUT -name "*hip_api_trace.csv" -size +30M -delete
