#!/bin/bash
# Round 2, GPU call M: kernel timeline of one bench step (gaps between kernels = host work and synchronisation inside the step).
set -u
export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out; LOG=gpurun_out/r2m.log; : > $LOG
rm -rf gpurun_out/prof_tl
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/prof_tl -o r -- python -u $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1 )
python - <<'PY' | tee -a $LOG
import csv, glob
ev = []
for fn in glob.glob("gpurun_out/prof_tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
for fn in glob.glob("gpurun_out/prof_tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
ev.sort()
# last step: find the last 'var_build_kernel' and go back to the first fb_kernel before it after previous commit
idx = [i for i, e in enumerate(ev) if e[2].startswith("commit_pad")]
end = idx[-1]; start = idx[-3] + 1 if len(idx) >= 3 else 0
t0 = ev[start][0]
prev_end = t0
busy = 0; gaps = []
for s, e, n in ev[start:end + 1]:
    if s - prev_end > 100000: gaps.append(((s - prev_end) / 1e6, (prev_end - t0) / 1e6, n))
    busy += (e - s)
    prev_end = max(prev_end, e)
print("step span %.1f ms, sum of event durations %.1f ms, events %d" % ((prev_end - t0) / 1e6, busy / 1e6, end + 1 - start))
for g in gaps: print("gap %.2f ms at t=%.1f ms before %s" % g)
print("sum of gaps > 0.1 ms: %.1f ms" % sum(g[0] for g in gaps))
PY
