#!/bin/bash
# GPU box: the m = 16 streaming row kernels with their stores batched per chunk of 32 tiles (k_phaseB_batched, k_phaseC_p0_batched:
# the default) against the plain kernels (BCG_ROW_BATCHED=0), alternating on one box: 64^4 and the capacity-mode share of 128^4.
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']; print('$1', round(d['ms_per_step'],3), {k: round(v/n,3) for k,v in d['kernel_ms'].items()})"; }
for b in 1 0 1 0 1 0; do
  BCG_ROW_BATCHED=$b python bench.py --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | line "64^4 BCG_ROW_BATCHED=$b"
done
for b in 1 0 1 0; do
  BCG_ROW_BATCHED=$b python bench.py --no-cpu-baseline --steps 8 --warmup 2 --local-dims 64 64 64 128 --capacity 32 2>/dev/null | line "64^3x128 ring 32 BCG_ROW_BATCHED=$b"
done
