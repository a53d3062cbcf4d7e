"""diag: rocprofv3 --sys-trace csv files -> the timeline (HIP API calls, copies, kernels) in a window around the n-th dispatch of a kernel.
usage: python diag/trace_window.py <dir> <kernel substring> <which> <ms before> <ms after>"""
import csv, glob, sys
d, name, which, before, after = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
f = lambda pat: glob.glob(d + "/**/*" + pat, recursive=True)[0]
k, a, m = f("kernel_trace.csv"), f("hip_api_trace.csv"), f("memory_copy_trace.csv")
ks = [r for r in csv.DictReader(open(k)) if name in r["Kernel_Name"]]
t = int(ks[which]["Start_Timestamp"])
rows = []
for path, kind, key in ((a, "API", "Function"), (m, "COPY", "Direction"), (k, "KERNEL", "Kernel_Name")):
    for r in csv.DictReader(open(path)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t - before * 1e6 < s < t + after * 1e6 and (e - s) > 3000:
            rows.append((s, kind, r.get(key, "")[:48], (e - s) / 1e6))
rows.sort()
for s, kind, nm, dur in rows:
    print("%9.3f ms  %-6s %-50s %.3f ms" % ((s - t) / 1e6, kind, nm, dur))
