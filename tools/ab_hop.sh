#!/bin/bash
# GPU box: stencil-only timing (tools/hop_only.py) plus L2->fabric read requests for several library builds.
# usage: tools/ab_hop.sh <variant> [<variant> ...]     ("base" = default build)
for v in "$@"; do
  if [ "$v" = base ]; then unset BCG_LIB; else export BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_$v.so; fi
  echo "== $v: $(python tools/hop_only.py 6 2>/dev/null)"
  PMC_PASSES=2 bash tools/pmc_passes.sh $v 2>&1 | grep -E "RDREQ_sum|TCC_HIT|TCC_MISS|TCC_READ" | awk '{print "   ", $1,$2,$3,$4,$5,$6,$7,$8,$9,$10}'
done
