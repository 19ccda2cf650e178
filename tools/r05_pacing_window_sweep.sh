#!/bin/bash
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']; print('$1', round(d['ms_per_step'],3), {k: round(v/n,3) for k,v in d['kernel_ms'].items() if 'hop' in k})"; }
for w in 4 1 2 3 6 8 4; do
  BCG_HOP_BUNDLE_SYNC=$w python bench.py --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | line "BUNDLE_SYNC=$w"
done
