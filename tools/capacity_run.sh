#!/bin/bash
# GPU-box script: capacity-mode checks.  1) the GPU test-suite, 2) bench at 64^4 in both modes, 3) the per-GPU share of 128^4
# (64x64x64x128 local sites, m=16, 4 shifts) on one MI355X in capacity mode, ring of 8 and of 16 slices.
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/cap_tests.log 2>&1 || { tail -30 gpurun_out/cap_tests.log; exit 1; }
tail -3 gpurun_out/cap_tests.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/cap_bench_default.json 2> gpurun_out/cap_bench_default.err &&
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --capacity 8 > gpurun_out/cap_bench_ring8.json 2> gpurun_out/cap_bench_ring8.err &&
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --local-dims 64 64 64 128 --capacity 8 > gpurun_out/cap_bench_128share_ring8.json 2> gpurun_out/cap_bench_128share.err &&
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --local-dims 64 64 64 128 --capacity 16 > gpurun_out/cap_bench_128share_ring16.json 2>> gpurun_out/cap_bench_128share.err
rc=$?
for f in gpurun_out/cap_bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step","iterations_per_sec","capacity_ring_slices","device_bytes_planned","device_bytes_in_use","device_bytes_total","kernel_ms","residual_after_timed_steps")})
except Exception as e:
    print("unreadable", e)
PY
done
tail -5 gpurun_out/cap_bench_128share.err
exit $rc
