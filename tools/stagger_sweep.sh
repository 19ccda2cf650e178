#!/bin/bash
# GPU box: does a per-field address stagger (BCG_FIELD_STAGGER, bytes) change the streaming kernels?  Fields of 64^4 x 768 B
# are exactly 12 GiB, allocated back to back: phase B / phase C read the same offset of 3 / 9 of them at once.
# usage: tools/stagger_sweep.sh <steps> <stagger> [<stagger> ...]
steps=$1; shift
for st in "$@"; do
  for rep in 1 2; do
    BCG_FIELD_STAGGER=$st python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); n=d['steps']
print('stagger $st', round(d['ms_per_step'],2), {k: round(x/n,2) for k,x in d['kernel_ms'].items()})"
  done
done
