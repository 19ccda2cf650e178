#!/usr/bin/env python3
"""GPU box: host <-> device rate of the boundary's layout-converting upload / download (bcg_field_upload/download),
for the PCIe-inclusive note in DESIGN.md section 6.  64^3 x 16 sites, m = 16 (3.2 GB per field); pageable host memory
(touched beforehand: first-touch page faults are the allocator's cost, not the transfer's) and pinned host memory
(bcg_host_alloc: what the C++ headers' operator[] mirror uses).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blockcg_amd as bc  # noqa: E402

dims, m = [64, 64, 64, 16], 16
ctx = bc.Context(dims)
f = bc.block_fermion_field(ctx, m).setRandom(seed=3)
g = bc.block_fermion_field(ctx, m)
out = {"sites": ctx.V, "m": m}
ref = None
for kind in ("pageable", "pinned"):
    t = time.perf_counter()
    h = np.empty((ctx.V, m, 3), dtype=np.complex128) if kind == "pageable" else f.pinned_array()
    h[:] = 1.0
    out[f"{kind}_alloc_and_touch_s"] = round(time.perf_counter() - t, 3)
    nbytes = h.nbytes
    for name, fn in (("download", lambda: f.download(out=h)), ("upload", lambda: g.upload(h))):
        fn()
        ctx.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            fn()
            ctx.synchronize()
            best = min(best, time.perf_counter() - t)
        out[f"{kind}_{name}_GBps"] = round(nbytes / best / 1e9, 1)
    # round trip is exact: g now holds what f holds
    if ref is None:
        ref = h[::4099].copy()
    assert np.array_equal(h[::4099], ref) and np.array_equal(g.download()[::4099], ref), kind
out["bytes"] = nbytes
print(json.dumps(out))
