#!/bin/bash
# GPU box: XCD patch shapes of the bundle sweep (p0 x p1 x p2 sites per x3 slice, 64 bundle tiles each), bench at 64^4, m = 16
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']; print('$1', round(d['ms_per_step'],3), {k: round(v/n,3) for k,v in d['kernel_ms'].items() if 'hop' in k})"; }
for p in 16,8,8 8,16,8 8,8,16 32,8,4 32,4,8 16,16,4 16,4,16 64,4,4 4,16,16 16,8,8; do
  BCG_HOP_PATCH=$p python bench.py --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | line "BCG_HOP_PATCH=$p"
done
