#!/bin/bash
out=gpurun_out/r04; mkdir -p $out
run() { echo -n "$1 " | tee -a $out/e3_hop.txt; shift; env "$@" timeout -k 10 300 python tools/hop_only.py 8 2>/dev/null | tee -a $out/e3_hop.txt; }
for rep in 1 2 3; do
  run pipe_paced X=1 || exit 1
  run pipe_unpaced BCG_HOP_BUNDLE_SYNC=0 || exit 1
  run pipe_paced8 BCG_HOP_BUNDLE_SYNC=8 || exit 1
  run nopipe_paced BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_nopipe.so || exit 1
done
export BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_pstamps.so
timeout -k 10 300 python tools/hop_stamps.py pipe 2>/dev/null | tee -a $out/e3_stamps.txt
