#!/usr/bin/env python3
"""GPU box: the field-level right-multiplications K5 / K6 (block_fermion_field.add / rescale_add with an m x m matrix,
inc/fields.hpp:70-90) at 64^4, m = 16 with batched stores (default) and plain (BCG_ROW_BATCHED=0), one process each, alternating."""
import os
import subprocess
import sys

code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import blockcg_amd as bc
dims, m = [64, 64, 64, 64], 16
ctx = bc.Context(dims)
x = bc.block_fermion_field(ctx, m).setRandom(seed=1)
y = bc.block_fermion_field(ctx, m).setRandom(seed=2)
rng = np.random.default_rng(3)
C = (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))) * 0.1
for _ in range(2):
    y.add(x, C); y.rescale_add(C, x, 1.0)
ctx.synchronize(); ctx.profiling(True); ctx.profile_reset()
for _ in range(10):
    y.add(x, C)
for _ in range(10):
    y.rescale_add(C, x, 1.0)
p = ctx.profile()
print("BCG_ROW_BATCHED=%s" % os.environ.get("BCG_ROW_BATCHED", "1"), {k: round(v["ms"] / v["count"], 3) for k, v in p.items() if k.startswith("block_")},
      "checksum %.15e" % float(np.abs(y.download_sites(np.arange(0, ctx.V, 65537))).sum()))
'''
for b in ("1", "0", "1", "0"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BCG_ROW_BATCHED=b), stderr=subprocess.DEVNULL)
