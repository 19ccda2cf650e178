#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs into profiles/hbm_traffic.json (bytes per launch per
kernel class), applying the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
counters are in KiB; FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores.
Usage: python tools/make_traffic_json.py <fetch_dir> <write_dir> <out.json> [<bench line .json of the profiled command>]
The optional bench line supplies the shape the counters were taken at ("_shape"), which bench.py checks before quoting them."""
import collections
import csv
import glob
import json
import os
import re
import sys

NAMES = [(r"k_phaseC_multi<\d+, \d+, 2[,>]", "phaseC_multi2"), (r"k_phaseC_multi<\d+, \d+, 3[,>]", "phaseC_multi3"),
         (r"k_phaseC_multi<\d+, \d+, 4[,>]", "phaseC_multi4"), (r"k_phaseC_p0(_batched)?<", "phaseC_p0"), (r"k_phaseC<", "phaseC"), (r"k_phaseB", "phaseB"), (r"k_hop4b<\d+, 0, false", "hop"), (r"k_hop4b<\d+, 1, true", "hop_shifted_gram"),
         (r"k_hop4b<\d+, 1, false", "hop_shifted"), (r"k_hop4c<\d+, 0, false", "hop"), (r"k_hop4c<\d+, 1, true", "hop_shifted_gram"),
         (r"k_hop4c<\d+, 1, false", "hop_shifted"), (r"k_hop4<\d+, 0, false", "hop"), (r"k_hop4<\d+, 1, true", "hop_shifted_gram"),
         (r"k_hop4<\d+, 1, false", "hop_shifted"), (r"k_hop_fast<\d+, 0", "hop"), (r"k_hop_fast<\d+, 1, true", "hop_shifted_gram")]


def per_kernel(d, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    # one pass per directory; gpurun merges every call's files back, so a local copy may hold older passes too: newest only
    for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for pat, name in NAMES:
                if re.search(pat, r["Kernel_Name"]):
                    agg[name][0] += float(r["Counter_Value"])
                    agg[name][1] += 1
                    break
    return {k: v[0] / v[1] for k, v in agg.items() if v[1]}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    rd = fetch.get(k, 0.0) * 1024 * 2
    wr = write.get(k, 0.0) * 1024
    out[k] = {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
              "source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and WRITE_SIZE, separate passes, averaged over launches"}
if len(sys.argv) > 4:
    b = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
    per_gpu = [g // p for g, p in zip(b["config"]["global_dims"], b["config"]["process_grid"])]
    out["_shape"] = {"local_dims": per_gpu, "m": b["config"]["m"], "n_shifts": len(b["config"]["shifts"]),
                     "capacity": b.get("capacity_ring_slices", 0),
                     "measured": "rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes of bench.py"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
