#!/bin/bash
# GPU box: a single system (S = 1) with its iterations grouped for the deferred X_0 update (default) against the plain form
# (BCG_PAIR_SHIFTS=0), alternating on one box: config 1 (32^4, m = 8) and 64^4, m = 16.
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']; print('$1', round(d['ms_per_step'],4), 'ms/iteration', {k: round(v/n,4) for k,v in d['kernel_ms'].items()})"; }
for rep in 1 2; do
  for pair in 4 0; do
    BCG_PAIR_SHIFTS=$pair python bench.py --no-cpu-baseline --steps 48 --warmup 8 --local-dims 32 32 32 32 --m 8 --shifts 1 2>/dev/null | line "32^4 m=8 S=1 BCG_PAIR_SHIFTS=$pair:"
  done
done
for rep in 1 2; do
  for pair in 4 0; do
    BCG_PAIR_SHIFTS=$pair python bench.py --no-cpu-baseline --steps 12 --warmup 4 --local-dims 64 64 64 64 --m 16 --shifts 1 2>/dev/null | line "64^4 m=16 S=1 BCG_PAIR_SHIFTS=$pair:"
  done
done
