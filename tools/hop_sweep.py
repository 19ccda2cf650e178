#!/usr/bin/env python3
"""Tuning aid (GPU box): time the stencil kernel at 64^4, m=16 for each tile-walk variant.
Usage: python tools/hop_sweep.py [reps]"""
import gc
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blockcg_amd as bc  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dims, m = [64, 64, 64, 64], 16
configs = [(3, "16,8,8", 512, 1), (3, "16,8,8", 768, 1), (3, "16,8,8", 768, 0), (3, "16,8,4", 768, 1), (3, "32,8,4", 768, 1), (3, "16,8,8", 1024, 1)]
for walk, patch, blocks, flags in configs:
    os.environ["BCG_HOP_WALK"] = str(walk)
    os.environ["BCG_HOP_PATCH"] = patch
    os.environ["BCG_HOP_BLOCKS"] = str(blocks)
    os.environ["BCG_HOP_FLAGS"] = str(flags)
    ctx = bc.Context(dims)
    D = bc.dirac_op(ctx, 0.1, seed=1)
    x = bc.block_fermion_field(ctx, m).setRandom(seed=2)
    y = bc.block_fermion_field(ctx, m)
    D.D(y, x)
    D.op(y, x)
    ctx.synchronize()
    ctx.profiling(True)
    ctx.profile_reset()
    for _ in range(reps):
        D.D(y, x)
    for _ in range(reps):
        D.op(y, x)
    prof = ctx.profile()
    hop_ms = prof["hop"]["ms"] / prof["hop"]["count"]
    hs = prof.get("hop_shifted", {"ms": 0, "count": 1})
    print(json.dumps({"walk": walk, "patch": patch, "blocks": blocks, "flags": flags, "hop_ms": round(hop_ms, 3),
                      "hop_shifted_ms": round(hs["ms"] / hs["count"], 3),
                      "hop_GBps_alg": round(ctx.V * (2 * 48 * m + 576) / hop_ms / 1e6, 1)}), flush=True)
    del x, y, D, ctx
    gc.collect()
