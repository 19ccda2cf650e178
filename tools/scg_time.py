#!/usr/bin/env python3
"""GPU box: per-iteration time of SCG (src/standard_solvers.cpp:34-95) with the shifts of the reference's benchmark driver,
32^4 sites, N_rhs = 1; the per-shift updates of an iteration are one launch (k_scg_update)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blockcg_amd as bc  # noqa: E402

dims = [32, 32, 32, 32]
shifts = [0, 0, 1e-10, 1e-8, 1e-6, 1e-5, 1e-4, 1e-2, 1e-1]  # benchmark.cpp:12-13
ctx = bc.Context(dims)
D = bc.dirac_op(ctx, 1e-3, seed=1)
b = bc.block_fermion_field(ctx, 1).setRandom(seed=2)
x = [bc.block_fermion_field(ctx, 1) for _ in shifts]
bc.SCG(x, b, D, shifts, 0.0, 0.0, max_iterations=20)
ctx.profiling(True)
ctx.profile_reset()
t = time.perf_counter()
it = bc.SCG(x, b, D, shifts, 0.0, 0.0, max_iterations=200)
ctx.synchronize()
dt = time.perf_counter() - t
prof = ctx.profile()
print(f"SCG 32^4, {len(shifts)} shifts: {dt / it * 1e3:.3f} ms per iteration;",
      {k: round(v["ms"] / it, 4) for k, v in prof.items() if v["ms"] > 0})
V = 32 ** 4
passes = 1 + 4 * len(shifts)
print(f"scg_update: {passes} field passes x {V * 48 / 1e6:.1f} MB = {passes * V * 48 / 1e9:.2f} GB per iteration")
