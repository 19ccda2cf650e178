#!/bin/bash
# GPU box: persistent-grid sizes of the fused row kernels (phase B, phase C; BCG_ROW_BLOCKS_B / _C, default 1024 = 4 blocks per CU)
# usage: tools/row_blocks_sweep.sh <steps> <blocks> [<blocks> ...]
steps=$1; shift
for nb in "$@"; do
  BCG_ROW_BLOCKS_B=$nb BCG_ROW_BLOCKS_C=$nb python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); n=d['steps']
print('row blocks $nb', round(d['ms_per_step'],2), {k: round(x/n,2) for k,x in d['kernel_ms'].items()})"
done
