#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`
in ROCm 7.x) into the per-kernel table `--stats` prints: calls, total/avg/min/max ms, percent.
usage: scripts/rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                           "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(workgroup_x) "
                           "from kernels group by name order by 3 desc"))
    tot = float(sum(r[2] for r in rows)) or 1.0
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPRs", "SGPRs", "LDS",
                "Scratch", "WorkgroupX"])
    for r in rows:
        w.writerow([r[0], r[1], int(r[2]), int(r[3]), "%.2f" % (100.0 * r[2] / tot), r[4], r[5], r[6], r[7], r[8], r[9], r[10]])


if __name__ == "__main__":
    main()
