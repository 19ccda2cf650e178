#!/bin/bash
# GPU box: parity of the stencil forms first (one bundle-sweep case alone, then the rest), then bench.py A/B between
# builds on the same device.   usage: tools/r04_pipe_ab.sh <tag> <variant> [<variant> ...]   ("base" = the default build)
tag=$1; shift
out=gpurun_out/r04
mkdir -p $out
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "column_sweep and 1-0-16-dims0" > $out/${tag}_first.log 2>&1 || { tail -30 $out/${tag}_first.log; exit 1; }
tail -1 $out/${tag}_first.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "column_sweep or specialised or capacity_mode_matches or fixed_work or cache_blocked or x3_carry" > $out/${tag}_parity.log 2>&1 || { tail -30 $out/${tag}_parity.log; exit 1; }
tail -2 $out/${tag}_parity.log
timeout -k 10 600 python -m pytest tests/test_fullsize_parity.py -x -q -m gpu -k "operator" > $out/${tag}_full.log 2>&1 || { tail -30 $out/${tag}_full.log; exit 1; }
tail -2 $out/${tag}_full.log
for rep in 1 2; do
  bash tools/ab_bench.sh 12 "$@" 2>&1 | tee -a $out/${tag}_ab.txt
done
