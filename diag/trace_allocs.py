"""diag: rocprofv3 --sys-trace csv -> every allocation / free / long API call after the first third of the run"""
import csv, glob, sys
d = sys.argv[1]
a = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(a))]
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t0 + (t1 - t0) // 3
from collections import Counter
cnt = Counter()
for s, e, f in rows:
    if s < cut: continue
    if "Malloc" in f or "Free" in f or (e - s) > 5_000_000:
        cnt[f] += 1
        if (e - s) > 1_000_000: print("%10.3f ms  %-28s %.3f ms" % ((s - t0) / 1e6, f, (e - s) / 1e6))
print(dict(cnt))
