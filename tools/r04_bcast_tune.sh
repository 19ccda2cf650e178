#!/bin/bash
# GPU box: the broadcast-link stencil (BCAST) under the other knobs -- spread masks (builds) and the pacing window (env) --
# tools/hop_only.py, 40 launches each, interleaved three times on one device.
out=gpurun_out/r04
mkdir -p $out
for rep in 1 2 3; do
  for v in base sp0 sp7 sp5; do
    if [ "$v" = base ]; then unset BCG_LIB; else export BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_$v.so; fi
    echo "$v $(timeout -k 10 120 python tools/hop_only.py 40 2>/dev/null | tail -1)" | tee -a $out/bcast_tune.txt
  done
  unset BCG_LIB
  for w in 3 6 8 12; do
    echo "window$w $(BCG_HOP_BUNDLE_SYNC=$w timeout -k 10 120 python tools/hop_only.py 40 2>/dev/null | tail -1)" | tee -a $out/bcast_tune.txt
  done
done
