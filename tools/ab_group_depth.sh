python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grouped or deferred" > gpurun_out/pair_test.log 2>&1; tail -3 gpurun_out/pair_test.log
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 12 > gpurun_out/ab_$name.json 2>gpurun_out/ab_$name.err; }
for i in 1 2; do
run d2_$i BCG_PAIR_SHIFTS=2
run d2nw8_$i BCG_PAIR_SHIFTS=2 BCG_PHASEC_MULTI_NW=8
run d3_$i BCG_PAIR_SHIFTS=3
run d4_$i BCG_PAIR_SHIFTS=4
done
python - <<EOF2
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_d*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f[14:-5], round(j["ms_per_step"],2), {k:round(v,1) for k,v in j["kernel_ms"].items()}, j["roofline"]["per_kernel_frac"], round(j["device_bytes_in_use"]/1e9,1))
    except Exception as e: print(f, "ERR", e)
EOF2
