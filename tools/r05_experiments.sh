#!/bin/bash
# GPU box, round 5: (A) the grid / prefetch variants of k_phaseC_p0, (B) the super-patch tile order of k_hop4b.  Alternating runs.
out=gpurun_out/r05
mkdir -p $out
{
echo "== A: k_phaseC_p0 (bench.py, 64^4 m=16 S=4; per-launch ms = total / launches)"
for spec in "BCG_P0_AHEAD=0 BCG_P0_BLOCKS=1024" "BCG_P0_AHEAD=1 BCG_P0_BLOCKS=1024" "BCG_P0_AHEAD=0 BCG_P0_BLOCKS=2048" "BCG_P0_AHEAD=1 BCG_P0_BLOCKS=512" "BCG_P0_AHEAD=0 BCG_P0_BLOCKS=1536" "BCG_P0_AHEAD=0 BCG_P0_BLOCKS=1024"; do
  echo "-- $spec: $(env $spec python bench.py --no-cpu-baseline --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']
print('ms/step %.2f' % d['ms_per_step'], 'p0 per launch %.3f' % (d['kernel_ms']['phaseC_p0']/(n*3/4)), {k: round(v/n,2) for k,v in d['kernel_ms'].items()})")"
done
echo "== B: super-patch order of k_hop4b (tools/hop_only.py 12: ms per launch)"
for rep in 1 2 3; do
  for s in 0 1; do echo "-- BCG_HOP_SUPER=$s: $(BCG_HOP_SUPER=$s python tools/hop_only.py 12 2>/dev/null | tail -1)"; done
done
echo "-- unpaced (BCG_HOP_BUNDLE_SYNC=0):"
for s in 0 1; do echo "-- BCG_HOP_SUPER=$s unpaced: $(BCG_HOP_BUNDLE_SYNC=0 BCG_HOP_SUPER=$s python tools/hop_only.py 12 2>/dev/null | tail -1)"; done
} > $out/experiments_1.txt 2>&1
cat $out/experiments_1.txt
