#!/usr/bin/env python3
"""GPU box: the long solve of tools/soak_solve.py (64^4, m = 16, 4 shifts, mass 0.05, eps 1e-10) in capacity mode (ring 16),
where the shift updates go two iterations at a time, X_0's with them in the spare-less form, and the closing phase B writes
over T: iterations, time, recomputed true residuals; then the same with BCG_PAIR_SHIFTS=0 (a second context) -- the solutions of
the shifted systems must be bit-identical, X_0 equal to rounding."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blockcg_amd as bc

dims, m, mass, eps = [64, 64, 64, 64], 16, 0.05, 1e-10
shifts = [0.0, 1e-6, 1e-4, 1e-2]
sample = np.arange(0, int(np.prod(dims)), 4099)
out = []
for pair in ("4", "0"):
    os.environ["BCG_PAIR_SHIFTS"] = pair
    ctx = bc.Context(dims)
    ctx.capacity_mode(16)
    D = bc.dirac_op(ctx, mass, seed=41)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=42)
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    t0 = time.time()
    it = bc.SBCGrQ(X, B, D, shifts, eps, eps, max_iterations=3000)
    dt = time.time() - t0
    res = bc.true_residuals(X, B, D, shifts)
    print("BCG_PAIR_SHIFTS=" + pair, "iterations", it, "seconds %.1f" % dt, "ms/iter %.2f" % (dt / it * 1e3),
          "max true residual per shift", res.max(axis=1), flush=True)
    out.append([x.download_sites(sample) for x in X])
    del X, B, D, ctx
print("sampled solutions of the shifts >= 1 bit-identical:", all(np.array_equal(a, b) for a, b in zip(out[0][1:], out[1][1:])),
      "| shift 0: max |dX| / max |X| = %.2e" % (np.abs(out[0][0] - out[1][0]).max() / np.abs(out[1][0]).max()))
