#!/bin/bash
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']; print('$1', round(d['ms_per_step'],2), 'ms/iteration', d.get('ring_chunks'), {k: round(v/n,2) for k,v in d['kernel_ms'].items()})"; }
for ch in 30 15 10 30; do
  BCG_RING_CHUNK=$ch python bench.py --no-cpu-baseline --steps 8 --warmup 2 --local-dims 64 64 64 128 --capacity 32 2>/dev/null | line "ring 32 BCG_RING_CHUNK=$ch:"
done
