#!/bin/bash
# GPU box: capacity-mode bench lines: 64^4 default vs ring of 8 slices, and the per-GPU share of 128^4 on one MI355X.
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/cap_bench_default.json 2> gpurun_out/cap_bench_default.err &&
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --capacity 8 > gpurun_out/cap_bench_ring8.json 2> gpurun_out/cap_bench_ring8.err &&
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --local-dims 64 64 64 128 --capacity 8 > gpurun_out/cap_bench_128share_ring8.json 2> gpurun_out/cap_bench_128share.err &&
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --local-dims 64 64 64 128 --capacity 16 > gpurun_out/cap_bench_128share_ring16.json 2>> gpurun_out/cap_bench_128share.err
rc=$?
for f in gpurun_out/cap_bench_default.json gpurun_out/cap_bench_ring8.json gpurun_out/cap_bench_128share_ring8.json gpurun_out/cap_bench_128share_ring16.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); n=d["steps"]
    print(round(d["value"]/1e6,1), "M site-iter/s", round(d["ms_per_step"],2), "ms", "ring", d["capacity_ring_slices"], "in use GB", round(d["device_bytes_in_use"]/1e9,1), {k: round(v/n,2) for k,v in d["kernel_ms"].items()})
except Exception as e:
    print("unreadable", e)
PY
done
exit $rc
