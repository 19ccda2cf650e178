#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 --pmc counter collections under a directory (tools/r04_bcast_pmc.sh):
for every <run>/…counter_collection.csv the mean counter value per launch of each k_hop4b instantiation."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    run = os.path.relpath(f, root).split(os.sep)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_hop4b" not in name:
            continue
        short = "plain" if "k_hop4b<16, 0" in name else "fused"
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(run, k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(acc[k].items())}, "launches", len(next(iter(acc[k].values()))))
