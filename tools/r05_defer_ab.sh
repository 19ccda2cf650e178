#!/bin/bash
# GPU box, round 5: deferred X_0 on / off, alternating on one box: bench.py (fixed work) and a converging solve (tools/soak_solve.py's first half)
out=gpurun_out/r05; mkdir -p $out
{
for rep in 1 2 3; do for d in 1 0; do
  echo "-- bench BCG_DEFER_X0=$d: $(BCG_DEFER_X0=$d python bench.py --no-cpu-baseline --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']
print('ms/step %.2f' % d['ms_per_step'], {k: round(v/n,2) for k,v in d['kernel_ms'].items()})")"
done; done
for rep in 1 2; do for d in 1 0; do
  echo "-- solve BCG_DEFER_X0=$d: $(BCG_DEFER_X0=$d python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import blockcg_amd as bc
dims, m, mass, eps = [64, 64, 64, 64], 16, 0.05, 1e-10
shifts = [0.0, 1e-6, 1e-4, 1e-2]
ctx = bc.Context(dims)
D = bc.dirac_op(ctx, mass, seed=41)
B = bc.block_fermion_field(ctx, m).setRandom(seed=42)
X = [bc.block_fermion_field(ctx, m) for _ in shifts]
st = bc.SBCGrQState(X, B, D, shifts, eps, eps)
st.iterate(8)
ctx.synchronize()
t0 = time.time()
it = st.iterate(3000)
ctx.synchronize()
dt = time.time() - t0
print("iterations", it, "ms/iter %.3f" % (dt / (it - 8) * 1e3))
PY
)"
done; done
} > $out/defer_ab.txt 2>&1
cat $out/defer_ab.txt
