#!/bin/bash
# Tuning aid (GPU box): bench.py against several builds of the library on the SAME device.
# usage: tools/ab_bench.sh <steps> <variant> [<variant> ...]   (variant "base" = the default build)
steps=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset BCG_LIB; else export BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_$v.so; fi
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); n=d['steps']
print('$v', round(d['ms_per_step'],2), round(d['iterations_per_sec'],3), {k: round(x/n,2) for k,x in d['kernel_ms'].items()})"
done
