#!/bin/bash
# Tuning aid (GPU box): per-kernel time and FETCH_SIZE (GB, x2-corrected) of bench.py for several library builds.
# usage: tools/ab_pmc.sh <variant> [<variant> ...]      ("base" = default build)
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset BCG_LIB; else export BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_$v.so; fi
  rm -rf gpurun_out/abpmc_$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/abpmc_$v -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/abpmc_$v.err
  python tools/pmc_summary.py gpurun_out/abpmc_$v | grep hop | sed "s/^/$v /"
done
