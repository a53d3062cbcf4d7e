#!/usr/bin/env python3
"""Generates tests/golden/msa_md5.json: MD5 of the final MSA the compiled reference
(oracle/_ref/muscle, built by oracle/build_ref.sh from /root/reference/src) writes for each named
input set of tests/_msa.py. Run in the container that has /root/reference; the JSON is committed."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _msa  # noqa: E402

SETS = ["n2_L40", "n3_L30", "n8_L60", "ragged", "bb11001", "bb11005", "n32_L150", "dupes", "consiters0", "perturb", "perturb_small",
        "synth_6x40_s2", "synth_5x1300_s3", "synth_3x1100_s3+r2", "synth_64x200_s1", "synth_128x300_s1", "super7_200x120_b32", "super7_8x18_b4", "super7_24x14_b3", "super5_14x20", "super5_120x80",
        "mega_bb11001", "mega_synth_6x40_s2", "mega_super7_12x30_b4", "mega_bb11001+r2", "mega_synth_6x40_s2+r2", "mega_super7_6x16_b3",
        "n2_L40+r2", "n3_L30+r2", "synth_6x40_s2+r2", "dupes+r2", "consiters0+r2", "perturb_small+r2",
        # BASELINE config 5 as stated: -super7 with -distmxin (reseek format, tests/_msa.py reseek_distmx); 2000 x 250 takes the
        # reference ~4 minutes on 7 threads, 10000 x 250 about an hour
        "super7dm_300x100_b16", "super7dm_2000x250_b32", "super7dm_10000x250_b32"]
out = {}
if "--all" not in sys.argv and os.path.exists(_msa.GOLDEN):  # default: only the sets that have no entry yet
    out = _msa.golden_md5()
for s in SETS:
    if s in out:
        continue
    out[s] = _msa.run_muscle(_msa.REF_MUSCLE, s, threads=os.cpu_count() or 4)[0]
    print(s, out[s])
with open(_msa.GOLDEN, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
