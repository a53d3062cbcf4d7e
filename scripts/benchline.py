#!/usr/bin/env python3
"""stdin: bench.py output (+ library trace lines) -> the few fields an A/B run is read for. usage: bench.py ... 2>&1 | python scripts/benchline.py"""
import json
import sys

for line in sys.stdin:
    line = line.rstrip()
    if line.startswith('{"metric"'):
        d = json.loads(line)
        k = d["kernel_ms_per_step"]
        print("pairs/s %.0f  ms/step %.1f  fb %.1f post %.1f store %.1f relax %.1f commit %.1f  parity %s" % (
            d["value"], d["ms_per_step"], k["fb"], k["post"], k["store_build"], k["relax"], k["commit"], d.get("parity_digest")))
        print("   " + d["relax_geometry"]["layout"][-330:])
    elif "timers" in line or "prefetched" in line or "worst step" in line or "WARNING" in line or "Error" in line or "error" in line:
        print(line[:400])
