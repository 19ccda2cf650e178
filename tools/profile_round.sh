#!/bin/bash
# GPU box: the measured artefacts of a round, from one build on one device:
#   gpurun_out/prof/bench.json            bench line (un-profiled)
#   gpurun_out/prof/stats/...             rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/prof/pmc_fetch, pmc_write  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)
#   gpurun_out/prof/hbm_traffic.json      bytes per launch per kernel class (tools/make_traffic_json.py)
# usage: tools/profile_round.sh
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/prof
rm -rf $out; mkdir -p $out
python bench.py --steps 20 --warmup 4 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err || { tail -5 $out/stats.err; exit 1; }
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python bench.py --steps 4 --warmup 4 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.err || { tail -5 $out/pmc_fetch.err; exit 1; }
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python bench.py --steps 4 --warmup 4 --no-cpu-baseline > /dev/null 2> $out/pmc_write.err || { tail -5 $out/pmc_write.err; exit 1; }
echo "write done"
python tools/make_traffic_json.py $out/pmc_fetch $out/pmc_write $out/hbm_traffic.json $out/bench_under_rocprof.json > /dev/null
(python tools/pmc_summary.py $out/pmc_fetch; python tools/pmc_summary.py $out/pmc_write) > $out/pmc_fetch_write.txt
cp $out/hbm_traffic.json profiles/hbm_traffic.json  # on the GPU box only: copy gpurun_out/prof/hbm_traffic.json into profiles/ after the call
python bench.py --steps 20 --warmup 4 > $out/bench_final.json 2>> $out/bench.err
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
python - <<'PY'
import json
d=json.loads(open("gpurun_out/prof/bench_final.json").read().strip().splitlines()[-1])
n=d["steps"]
print(json.dumps({k:d[k] for k in ("value","ms_per_step","iterations_per_sec","hbm_GBps_algorithmic","hbm_GBps_moved_per_gpu","hbm_GBps_pmc_per_gpu","roofline","cpu_baseline")}, indent=1))
print({k: round(v/n,2) for k,v in d["kernel_ms"].items()})
PY
