import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import blockcg_amd as bc
ring = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # > 0: capacity mode with that ring (groups of two, the spare-less form)
dims, m, mass, eps = [16, 16, 16, 16], 16, 1e-3, 1e-10
shifts = [0.0, 1e-6, 1e-4, 1e-2]
out = {}
for d in ("1", "0"):
    os.environ["BCG_DEFER_X0"] = d
    ctx = bc.Context(dims)
    if ring:
        ctx.capacity_mode(ring)
    D = bc.dirac_op(ctx, mass, seed=41)
    B = bc.block_fermion_field(ctx, m).setRandom(seed=42)
    X = [bc.block_fermion_field(ctx, m) for _ in shifts]
    it = bc.SBCGrQ(X, B, D, shifts, eps, eps, max_iterations=20000)
    res = bc.true_residuals(X, B, D, shifts)
    out[d] = (it, res.max(axis=1), [x.download() for x in X])
    print("BCG_DEFER_X0=" + d, "iterations", it, "max true residual per shift", res.max(axis=1))
    del X, B, D
    ctx.close()
for s in range(len(shifts)):
    a, b = out["1"][2][s], out["0"][2][s]
    print("shift", s, "|X_deferred - X_plain| / |X_plain| = %.3e" % (np.linalg.norm(a - b) / np.linalg.norm(b)), "bit-identical" if np.array_equal(a, b) else "")
