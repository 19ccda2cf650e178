#!/bin/bash
# GPU box: what the x3 chunks of the overlapped half-volume operator cost, on one GPU (no exchanges): bench.py --half at the
# ladder's share 128 x 64 x 32 x 128 with whole sweeps and with chunks of 32 / 16 / 8 slices (BCG_HALF_CHUNK_FORCE=1).
out=gpurun_out/r04; mkdir -p $out
for chunk in 0 32 16 8; do
  if [ $chunk = 0 ]; then unset BCG_HALF_CHUNK_FORCE BCG_HALF_CHUNK; else export BCG_HALF_CHUNK_FORCE=1 BCG_HALF_CHUNK=$chunk; fi
  timeout -k 10 300 python bench.py --half --local-dims 128 64 32 128 --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['steps']
print('chunk $chunk:', round(d['ms_per_step'],2), 'ms per step', round(d['device_bytes_in_use']/1e9,1), 'GB', {k: round(v/n,2) for k,v in d['kernel_ms'].items()}, d['stencil_kernel_launches'])" | tee -a $out/half_chunk.txt
done
