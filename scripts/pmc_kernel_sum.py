#!/usr/bin/env python3
"""usage: pmc_kernel_sum.py <dir with *counter_collection.csv> <kernel name substring>: per counter, the sum over the matching
kernel's dispatches and the dispatch count (A/B runs of one kernel under different knobs)"""
import csv
import glob
import sys

d, sub = sys.argv[1], sys.argv[2]
tot, cnt = {}, {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0) + 1
for k in sorted(tot):
    print("%-24s %.4e  (%d dispatches)" % (k, tot[k], cnt[k]))
