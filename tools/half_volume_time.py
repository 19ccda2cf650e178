#!/usr/bin/env python3
"""GPU box: per-iteration time and device memory of a half-volume SBCGrQ solve (one parity) next to the full-volume one,
64^4 (default) or the lattice given, m = 16, 4 shifts, fixed work.  Prints one JSON line (profiles/r03_half_volume.json,
profiles/r04_half_volume_128share.json)."""
import json
import os
import sys
import time

import torch  # noqa: F401  (one HIP runtime: see blockcg_amd/_lib.py)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blockcg_amd as bc  # noqa: E402

# usage: half_volume_time.py [L0 L1 L2 L3 [capacity-ring for the full-volume solve]]
#   64 64 64 128 32 = the per-GPU share of 128^4: one parity of it (whole tmp, shift updates grouped over four iterations)
#   next to the full-volume solve in capacity mode (ring 32, groups of two)
args = [int(a) for a in sys.argv[1:]]
dims = args[:4] if len(args) >= 4 else [64, 64, 64, 64]
ring = args[4] if len(args) >= 5 else 0
m, shifts, mass = 16, [0.0, 1e-6, 1e-4, 1e-2], 1e-3
out = {"dims": dims, "m": m, "shifts": len(shifts), "capacity_ring_of_the_full_solve": ring}
for name, parity in (("half", 0), ("full", None)):
    ctx = bc.Context(dims)
    if parity is None and ring:
        ctx.capacity_mode(ring)
    D = bc.dirac_op(ctx, mass, seed=1)
    B = bc.block_fermion_field(ctx, m, parity=parity).setRandom(seed=2)
    X = [bc.block_fermion_field(ctx, m, parity=parity) for _ in shifts]
    st = bc.SBCGrQState(X, B, D, shifts, 0.0, 0.0, consume_B=True)
    st.iterate(4)
    free, total = torch.cuda.mem_get_info(0)
    ctx.profiling(True)
    ctx.profile_reset()
    ctx.synchronize()
    t = time.perf_counter()
    st.iterate(8)
    ctx.synchronize()
    dt = (time.perf_counter() - t) / 8
    prof = ctx.profile()
    out[name] = {"ms_per_iteration": round(dt * 1e3, 2), "device_GB_in_use": round((total - free) / 1e9, 1),
                 "kernel_ms": {k: round(v["ms"] / 8, 2) for k, v in prof.items() if not k.startswith("stencil_form_")}}
    st.end()
    del st, X, B, D
    ctx.close()
    del ctx
out["two_half_solves_over_one_full"] = round(2 * out["half"]["ms_per_iteration"] / out["full"]["ms_per_iteration"], 3)
print(json.dumps(out))
