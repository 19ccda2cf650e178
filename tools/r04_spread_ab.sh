#!/bin/bash
# GPU box: the two stencil kernels alone (tools/hop_only.py, 40 launches each), builds interleaved three times on one device.
# usage: tools/r04_spread_ab.sh <variant> [<variant> ...]   ("base" = the default build)
out=gpurun_out/r04
mkdir -p $out
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset BCG_LIB; else export BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_$v.so; fi
    echo "$v $(timeout -k 10 120 python tools/hop_only.py 40 2>/dev/null | tail -1)" | tee -a $out/spread_hop_only.txt
  done
done
