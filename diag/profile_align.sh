#!/bin/bash
# diag/profile_align.sh <tag> [N L] — rocprofv3 kernel statistics and one SQ counter pass of `muscle_gpu -align` on the synthetic
# family (the join kernels: build_post_*, calc_aln_*), written under gpurun_out/<tag>_align_*; run on the GPU box.
set -u
TAG=${1:?tag}; N=${2:-1000}; L=${3:-400}
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R && python -c "from muscle_amd.synth import make_family, write_fasta; write_fasta('/tmp/in_${N}x${L}.fa', make_family($N, $L, seed=1))"
cd /tmp
B=$R/hostcxx/_build/muscle_gpu
rm -rf $OUT/${TAG}_align_stats $OUT/${TAG}_align_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_align_stats -o r -- $B -align /tmp/in_${N}x${L}.fa -output /tmp/o1.afa -threads 16 -quiet 2>&1 | tail -2
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/${TAG}_align_sq -o r -- $B -align /tmp/in_${N}x${L}.fa -output /tmp/o2.afa -threads 16 -quiet 2>&1 | tail -2
cp $OUT/${TAG}_align_stats/r_kernel_stats.csv $OUT/${TAG}_align_kernel_stats_${N}x${L}.csv
python - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open("$OUT/${TAG}_align_sq/r_counter_collection.csv")):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
with open("$OUT/${TAG}_align_sq_summary_${N}x${L}.csv", "w") as f:
    names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"]
    f.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(agg, key=lambda k: -agg[k]["SQ_WAVE_CYCLES"]):
        f.write(k + "," + str(cnt[k]) + "," + ",".join("%.6g" % agg[k][n] for n in names) + "\n")
print(open("$OUT/${TAG}_align_sq_summary_${N}x${L}.csv").read()[:3000])
PY
find $OUT/${TAG}_align_stats $OUT/${TAG}_align_sq -name "*kernel_trace.csv" -size +20M -delete
find $OUT/${TAG}_align_sq -name "*counter_collection.csv" -size +30M -delete
head -16 $OUT/${TAG}_align_kernel_stats_${N}x${L}.csv | cut -c1-200
