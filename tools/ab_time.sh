#!/bin/bash
# GPU box: stencil-only timing (tools/hop_only.py) for library variants x environment settings.
# usage: tools/ab_time.sh "<variant> [ENV=VAL ...]" ...      ("base" = default build)
for spec in "$@"; do
  set -- $spec; v=$1; shift
  if [ "$v" = base ]; then unset BCG_LIB; else export BCG_LIB=$PWD/blockcg_amd/_build/libblockcg_hip_$v.so; fi
  echo "== $spec: $(env "$@" python tools/hop_only.py 6 2>/dev/null | tail -1 | cut -c1-60)"
done
