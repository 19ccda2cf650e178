#!/bin/bash
# GPU box: the 64^3 x 128 share of 128^4 in capacity mode (ring 32) with X_0 deferred in the spare-less form (default) and
# without (BCG_DEFER_X0=0), alternating on one box; then ring 8.
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']; print('$1', round(d['ms_per_step'],2), 'ms/iteration', round(d['device_bytes_in_use']/1e9,1), 'GB', {k: round(v/n,2) for k,v in d['kernel_ms'].items()})"; }
for rep in 1 2; do
  for x in 1 0; do
    BCG_DEFER_X0=$x python bench.py --no-cpu-baseline --steps 8 --warmup 2 --local-dims 64 64 64 128 --capacity 32 2>/dev/null | line "ring 32 BCG_DEFER_X0=$x:"
  done
done
for x in 1 0; do
  BCG_DEFER_X0=$x python bench.py --no-cpu-baseline --steps 8 --warmup 2 --local-dims 64 64 64 128 --capacity 8 2>/dev/null | line "ring 8 BCG_DEFER_X0=$x:"
done
