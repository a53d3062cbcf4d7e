#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into profiles/pmc_traffic.json.
usage: scripts/pmc_summary.py N LEN fetch_dir write_dir [out.json [sq_dir ...]]   (LEN 0 = the rdrp input)
Each dir holds the csv output of `rocprofv3 --pmc FETCH_SIZE` (resp. WRITE_SIZE) `--output-format csv`.
HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md §HBM (FETCH_SIZE on gfx950
reports half the bytes of wide coalesced reads; both counters are in KiB); per launch = mean over
the dispatches of the kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(dirname, counter):
    out = defaultdict(list)
    for fn in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter:
                    out[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return out


def main():
    n, length, fdir, wdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                               "profiles", "pmc_traffic.json")
    fetch, write = read(fdir, "FETCH_SIZE"), read(wdir, "WRITE_SIZE")
    try:
        with open(out) as f:
            d = json.load(f)
    except (OSError, ValueError):
        d = {}
    # several instantiations of one template in a run (relax_var_kernel's two geometries on real data): the entry describes the
    # one that moved the most data
    def short_of(kname):
        return kname.split("(")[0].replace("void ", "").split("<")[0]
    weight = {k: sum(fetch.get(k, [])) + sum(write.get(k, [])) for k in set(fetch) | set(write)}
    dominant = {}
    for k, w in weight.items():
        if short_of(k) not in dominant or w > weight[dominant[short_of(k)]]:
            dominant[short_of(k)] = k
    for kname in sorted(dominant.values()):
        short = short_of(kname)
        fv, wv = fetch.get(kname, []), write.get(kname, [])
        if not fv and not wv:
            continue
        fm = sum(fv) / len(fv) if fv else 0.0
        wm = sum(wv) / len(wv) if wv else 0.0
        d["%s@%dx%d" % (short, n, length)] = {
            "kernel": kname, "launches_sampled": max(len(fv), len(wv)), "FETCH_SIZE_KiB_mean": fm, "WRITE_SIZE_KiB_mean": wm,
            "hbm_bytes_per_launch": (2.0 * fm + wm) * 1024.0,
            "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950)"}
        if os.environ.get("PMC_SOURCE"):
            d["%s@%dx%d" % (short, n, length)]["source"] = os.environ["PMC_SOURCE"]
        print(short, d["%s@%dx%d" % (short, n, length)])
    # SQ passes next to the FETCH/WRITE directories (prof_sq, prof_sq2), when present: instruction counts per launch
    base = os.path.dirname(os.path.normpath(fdir))
    sq = defaultdict(lambda: defaultdict(list))
    sq_dirs = sys.argv[6:] if len(sys.argv) > 6 else [os.path.join(base, "prof_sq"), os.path.join(base, "prof_sq2")]
    for sub in sq_dirs:
        for fn in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
            with open(fn) as f:
                for row in csv.DictReader(f):
                    sq[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for kname, counters in sq.items():
        short = short_of(kname)
        key = "%s@%dx%d" % (short, n, length)
        if key in d and d[key]["kernel"] == kname:
            d[key]["sq_per_launch"] = {c: sum(v) / len(v) for c, v in sorted(counters.items())}
    # PMC_STATS=<kernel_stats.csv of `rocprofv3 --kernel-trace --stats` of the same command>: the launch duration under the profiler,
    # and with it the clock the chip sustained under THIS kernel: SQ_BUSY_CYCLES is counted once per shader engine (MI355X: 8 XCDs
    # x 4 = 32), each busy for the whole launch of a grid that fills the chip, so clock = SQ_BUSY_CYCLES / 32 / duration.
    # PMC_GRBM=<dir of a --pmc GRBM_GUI_ACTIVE pass>: the same from the graphics register bus manager's busy counter (one per XCD).
    stats = os.environ.get("PMC_STATS")
    if stats and os.path.exists(stats):
        with open(stats) as f:
            for row in csv.DictReader(f):
                key = "%s@%dx%d" % (short_of(row["Name"]), n, length)
                if key in d and d[key]["kernel"] == row["Name"]:
                    dur = float(row["AverageNs"]) * 1e-9
                    d[key]["avg_duration_s_profiled"] = dur
                    busy = d[key].get("sq_per_launch", {}).get("SQ_BUSY_CYCLES")
                    if busy and dur > 0:
                        d[key]["clock_hz"] = busy / 32.0 / dur
                        d[key]["clock_method"] = "SQ_BUSY_CYCLES (one count per shader engine, 32 on MI355X) / 32 / mean launch duration of the --kernel-trace --stats pass"
    grbm = os.environ.get("PMC_GRBM")
    if grbm:
        for kname, vals in read(grbm, "GRBM_GUI_ACTIVE").items():
            key = "%s@%dx%d" % (short_of(kname), n, length)
            if key in d and d[key]["kernel"] == kname and vals:
                d[key]["GRBM_GUI_ACTIVE_per_launch"] = sum(vals) / len(vals)
                dur = d[key].get("avg_duration_s_profiled")
                if dur:
                    d[key]["clock_hz_grbm_if_one_counter_per_xcd"] = d[key]["GRBM_GUI_ACTIVE_per_launch"] / 8.0 / dur
    with open(out, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
