#!/bin/bash
# diag/align_back_to_back.sh — `muscle_gpu -align` 1000 x L~400 several times in a row, without and with a pause between the runs:
# does a run pay for the release of the previous process's device memory? usage: bash diag/align_back_to_back.sh
R=${GRAFT_REPO_ROOT:-$PWD}
python - <<PY
import sys; sys.path.insert(0, "$R")
from muscle_amd.synth import make_family, write_fasta
write_fasta("/tmp/in1000.fa", make_family(1000, 400, seed=1))
PY
cd /tmp
one() { local t0=$(date +%s%N); MUSCLE_GPU_TIMING=1 "$@" $R/hostcxx/_build/muscle_gpu -align /tmp/in1000.fa -output /tmp/o.afa -threads 16 -quiet 2>&1 | grep -E "CalcPosteriors \(replaced\)" | cut -c1-60; echo "  wall $(( ($(date +%s%N) - t0) / 1000000 )) ms  ($*)"; }
echo "--- back to back"; one; one; one
echo "--- 4 s pause before each"; sleep 4; one; sleep 4; one
for g in ${SCRATCH_LIST:-4 8}; do echo "--- back to back, $g GB of scratch instead of 16"; one env MPCGPU_SCRATCH_GB=$g; one env MPCGPU_SCRATCH_GB=$g; one env MPCGPU_SCRATCH_GB=$g; done
md5sum /tmp/o.afa
