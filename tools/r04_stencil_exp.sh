#!/bin/bash
# GPU box: stencil-only timing of several builds on the same device (tools/hop_only.py: hop and hop+Gram at 64^4, m = 16),
# then the per-phase stamps of the stamp builds.   usage: tools/r04_stencil_exp.sh <tag> "<timing variants>" "<stamp variants>"
tag=$1; out=gpurun_out/r04; mkdir -p $out
for rep in 1 2 3; do
  for v in $2; do
    if [ "$v" = base ]; then unset BCG_LIB; else export BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_$v.so; fi
    echo -n "$v " | tee -a $out/${tag}_hop.txt
    timeout -k 10 300 python tools/hop_only.py 8 2>/dev/null | tee -a $out/${tag}_hop.txt || exit 1
  done
done
for v in $3; do
  export BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_$v.so
  echo "== stamps $v" | tee -a $out/${tag}_stamps.txt
  BCG_HOP_BUNDLE_SYNC=0 timeout -k 10 300 python tools/hop_stamps.py pipe 2>/dev/null | tee -a $out/${tag}_stamps.txt || exit 1
done
