#!/bin/bash
# GPU box rehearsal: bench.py with 2 ranks sharing GPU 0 over gloo and an explicit process grid (absolute times are
# meaningless -- two processes share the card and the halo goes through the host -- the split of the stencil time
# between the interior and the boundary launch is what this shows).   usage: tools/two_rank.sh "1,1,2,1" [local dims]
grid=$1; shift
dims=${@:-64 64 64 32}
BCG_BENCH_GRID=$grid BCG_BACKEND=gloo BCG_DEVICE=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 1 --local-dims $dims 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']
print('$grid', {k: round(v/n,2) for k,v in d['kernel_ms'].items() if k.startswith('hop') or k.startswith('stencil')}, 'residual', d['residual_after_timed_steps'])"
