#!/usr/bin/env python3
"""GPU box: a long full-size solve (64^4, m = 16, 4 shifts, mass 0.05, eps 1e-10) with the true residuals recomputed
independently, then the same solve as two half-volume solves.  Last run (profiles/r03_soak_solve.txt): 954 iterations in
49.1 s (51.5 ms each), max true residual 9.9e-11 on every shift; half-volume: 952 + 952 iterations in 53.5 s, same residuals."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blockcg_amd as bc
dims, m, mass, eps = [64, 64, 64, 64], 16, 0.05, 1e-10
shifts = [0.0, 1e-6, 1e-4, 1e-2]
ctx = bc.Context(dims)
D = bc.dirac_op(ctx, mass, seed=41)
B = bc.block_fermion_field(ctx, m).setRandom(seed=42)
X = [bc.block_fermion_field(ctx, m) for _ in shifts]
t0 = time.time()
it = bc.SBCGrQ(X, B, D, shifts, eps, eps, max_iterations=3000)
dt = time.time() - t0
res = bc.true_residuals(X, B, D, shifts)
print("iterations", it, "seconds %.1f" % dt, "ms/iter %.2f" % (dt / it * 1e3), "max true residual per shift", res.max(axis=1))
# the same solve as two half-volume solves (one per site parity): iterations per parity, time, residuals on the full lattice
X2 = [bc.block_fermion_field(ctx, m) for _ in shifts]
t0 = time.time()
its = bc.SBCGrQ_half_volume(X2, B, D, shifts, eps, eps, max_iterations=3000)
dt2 = time.time() - t0
res2 = bc.true_residuals(X2, B, D, shifts)
diff = max(float(np.abs(a.download_sites(np.arange(0, ctx.V, 65537)) - b.download_sites(np.arange(0, ctx.V, 65537))).max()) for a, b in zip(X, X2))
print("half-volume: iterations (even, odd)", its, "seconds %.1f" % dt2, "max true residual per shift", res2.max(axis=1),
      "max |X_full - X_half| at sampled sites %.2e" % diff)
