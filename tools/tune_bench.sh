#!/bin/bash
# Tuning aid (GPU box): bench.py under different stencil tuning overrides on the SAME device.
# usage: tools/tune_bench.sh <steps> "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]
steps=$1; shift
for cfg in "$@"; do
  env $cfg python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); n=d['steps']
print('$cfg', round(d['ms_per_step'],2), round(d['iterations_per_sec'],3), {k: round(x/n,2) for k,x in d['kernel_ms'].items() if k.startswith('hop') or k.startswith('phase')})"
done
