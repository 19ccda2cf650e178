#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: python tools/pmc_summary.py <dir> [<dir> ...]
(the newest pass of each directory only: gpurun merges every call's files back, so a local copy may hold older passes too)"""
import collections
import csv
import glob
import os
import re
import sys

for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            k = (m.group(1) if m else r["Kernel_Name"][:30], r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
            agg[k][2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        for k, v in sorted(agg.items()):
            if any(x in k[0] for x in ("hop", "phase")):
                print("%-28s %-30s avg/launch %.5g  n=%d avg_ms=%.2f" % (k[0], k[1], v[0] / v[1], v[1], v[2] / v[1]))
