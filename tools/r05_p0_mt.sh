#!/bin/bash
# GPU box, round 5: k_phaseC_p0 with contiguous tile moves, ownership changed on the matrix pipe (BCG_P0_MT: 0 off, 1 loads, 2 + store)
out=gpurun_out/r05; mkdir -p $out
{
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "grouped_shift_updates_are_bit_identical and m16" 2>&1 | tail -2
for mt in 1 2; do echo "parity with BCG_P0_MT=$mt: $(BCG_P0_MT=$mt python -m pytest tests/test_gpu_parity.py -q -m gpu -k 'grouped_shift_updates_are_bit_identical and m16' 2>&1 | tail -1)"; done
for spec in "BCG_P0_MT=0" "BCG_P0_MT=1" "BCG_P0_MT=2" "BCG_P0_MT=2 BCG_P0_BLOCKS=1024" "BCG_P0_MT=2 BCG_P0_AHEAD=0 BCG_P0_BLOCKS=1024" "BCG_P0_MT=1 BCG_P0_BLOCKS=1024" "BCG_P0_MT=0" "BCG_P0_MT=2"; do
  echo "-- $spec: $(env $spec python bench.py --no-cpu-baseline --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']
print('ms/step %.2f' % d['ms_per_step'], 'p0 per launch %.3f' % (d['kernel_ms']['phaseC_p0']/(n*3/4)), {k: round(v/n,2) for k,v in d['kernel_ms'].items()})")"
done
} > $out/p0_mt.txt 2>&1
cat $out/p0_mt.txt
