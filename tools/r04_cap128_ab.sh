#!/bin/bash
# GPU box: the 128^4 share (64^3 x 128, ring 32) with the stencil's pacing forced on / the round-3 step, same device
out=gpurun_out/r04; mkdir -p $out
run() { echo -n "$1 " | tee -a $out/cap128_ab.txt; shift; env "$@" python bench.py --no-cpu-baseline --steps 4 --warmup 2 --local-dims 64 64 64 128 --capacity 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); n=d['steps']; print(round(d['ms_per_step'],2), {k: round(v/n,2) for k,v in d['kernel_ms'].items()})" | tee -a $out/cap128_ab.txt; }
for rep in 1 2; do
  run default X=1
  run paced4 BCG_HOP_BUNDLE_SYNC=-4
  run nopipe BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_nopipe.so
done
