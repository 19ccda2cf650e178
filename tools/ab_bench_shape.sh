#!/bin/bash
# GPU box: bench.py (no CPU leg) at a given shape under environment settings:
#   tools/ab_bench_shape.sh "<bench args>" "ENV=VAL ..." ...
args=$1; shift
for spec in "$@"; do
  echo "== $spec: $(env $spec python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']
print('ms/step %.3f' % d['ms_per_step'], 'sum kernels %.3f' % (sum(d['kernel_ms'].values())/n), {k: round(v/n,3) for k,v in d['kernel_ms'].items()})")"
done
