#!/bin/bash
# GPU box: time and fabric traffic of the two stencil kernels (tools/hop_only.py) for library variants x environment
# settings; FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes.
# usage: tools/ab_fetch.sh "<variant> [ENV=VAL ...]" ...      ("default" = the default build)
# prints per spec: plain ms, then per kernel FETCH_SIZE x 2 + WRITE_SIZE in GB (the gfx950 correction of FETCH_SIZE)
export TMPDIR=/tmp
n=0
for spec in "$@"; do
  set -- $spec; v=$1; shift
  n=$((n+1))
  if [ "$v" = default ]; then unset BCG_LIB; else export BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_$v.so; fi
  echo "== $spec"
  echo "   plain: $(env "$@" python tools/hop_only.py 6 2>/dev/null | tail -1 | cut -c1-70)"
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/abf_${n}_$c
    rm -rf $d
    env "$@" rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python tools/hop_only.py 4 > $d.out 2> $d.err || { echo "   $c failed"; tail -3 $d.err; continue; }
    python tools/pmc_summary.py $d | grep "k_hop4" | sed "s/^/   /"
  done
done
