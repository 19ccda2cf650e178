import os, sys, time, subprocess
for b in ("1", "0", "1", "0"):
    env = dict(os.environ, BCG_ROW_BATCHED=b)
    code = """
import os, sys, time
sys.path.insert(0, os.getcwd())
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
print("BCG_ROW_BATCHED=%s" % os.environ["BCG_ROW_BATCHED"], "iterations", it, "seconds %.1f" % dt, "ms/iter %.2f" % (dt / it * 1e3), "max true residual per shift", res.max(axis=1), flush=True)
"""
    subprocess.run([sys.executable, "-c", code], env=env, stderr=subprocess.DEVNULL)
