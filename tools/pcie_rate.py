#!/usr/bin/env python3
"""GPU box: host <-> device rate of the boundary's layout-converting upload / download (bcg_field_upload/download),
for the PCIe-inclusive note in DESIGN.md section 6.  64^3 x 16 sites, m = 16 (3.2 GB per field)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blockcg_amd as bc  # noqa: E402

dims, m = [64, 64, 64, 16], 16
ctx = bc.Context(dims)
f = bc.block_fermion_field(ctx, m)
h = np.zeros((ctx.V, m, 3), dtype=np.complex128)
h[:] = 1.0
nbytes = h.nbytes
for name, fn in (("upload", lambda: f.upload(h)), ("download", lambda: f.download())):
    fn()
    ctx.synchronize()
    t = time.perf_counter()
    fn()
    ctx.synchronize()
    dt = time.perf_counter() - t
    print(f"{name}: {nbytes / 1e9:.2f} GB in {dt * 1e3:.1f} ms = {nbytes / dt / 1e9:.1f} GB/s (pageable host memory, staged, layout-converted)")
