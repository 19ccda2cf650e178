#!/bin/bash
# GPU box: parity tests with the direction-split stencil forced on, then its timing / L2 fetches against the default.
set -o pipefail
mkdir -p gpurun_out
BCG_HOP_SPLIT=1 python -m pytest tests/test_gpu_parity.py tests/test_distributed_gpu.py -m gpu -x -q -k "specialised or capacity or fixed_work or carry or domain_decomposed or config2 or mfma_fast" > gpurun_out/split_tests.log 2>&1 || { tail -30 gpurun_out/split_tests.log; exit 1; }
tail -2 gpurun_out/split_tests.log
for cfg in "0 8,8,8" "1 8,8,8" "1 16,8,4" "1 8,8,4" "1 8,16,8"; do
  set -- $cfg
  echo "#### split=$1 patch_split=$2"
  BCG_HOP_SPLIT=$1 BCG_HOP_PATCH_SPLIT=$2 bash tools/shape_fetch.sh "64 64 64 64"
done
