#!/bin/bash
# GPU box: bench.py (no CPU leg) under environment settings: tools/ab_bench_env.sh "ENV=VAL ..." ...
for spec in "$@"; do
  echo "== $spec: $(env $spec python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']
print('ms/step %.2f' % d['ms_per_step'], {k: round(v/n,2) for k,v in d['kernel_ms'].items()})")"
done
