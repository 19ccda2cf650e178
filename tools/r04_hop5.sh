#!/bin/bash
# GPU box: k_hop5 -- one small case first, then the stencil parity tests, then timing against k_hop4b (BCG_HOP5=0)
out=gpurun_out/r04; mkdir -p $out; tag=$1
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "column_sweep and 1-0-16-dims0" > $out/${tag}_first.log 2>&1 || { tail -30 $out/${tag}_first.log; exit 1; }
tail -1 $out/${tag}_first.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "column_sweep or specialised or capacity_mode_matches or fixed_work or cache_blocked or x3_carry or true_residual" > $out/${tag}_parity.log 2>&1 || { tail -30 $out/${tag}_parity.log; exit 1; }
tail -2 $out/${tag}_parity.log
timeout -k 10 600 python -m pytest tests/test_fullsize_parity.py -x -q -m gpu -k "operator" > $out/${tag}_full.log 2>&1 || { tail -30 $out/${tag}_full.log; exit 1; }
tail -2 $out/${tag}_full.log
run() { echo -n "$1 " | tee -a $out/${tag}_hop.txt; shift; env "$@" timeout -k 10 300 python tools/hop_only.py 8 2>/dev/null | tee -a $out/${tag}_hop.txt; }
for rep in 1 2 3; do
  run hop5 X=1 || exit 1
  run hop4b BCG_HOP5=0 || exit 1
done
for rep in 1 2; do
  echo -n "bench hop5 " | tee -a $out/${tag}_hop.txt; python bench.py --no-cpu-baseline --steps 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v/d['steps'],2) for k,v in d['kernel_ms'].items()}, d['stencil_kernel_launches'])" | tee -a $out/${tag}_hop.txt
  echo -n "bench hop4b " | tee -a $out/${tag}_hop.txt; BCG_HOP5=0 python bench.py --no-cpu-baseline --steps 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k: round(v/d['steps'],2) for k,v in d['kernel_ms'].items()}, d['stencil_kernel_launches'])" | tee -a $out/${tag}_hop.txt
done
