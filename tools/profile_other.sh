#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the secondary shapes and of the capacity-mode 128^4 share
# -> gpurun_out/prof_other/{m8,m32,cap128}_kernel_stats.csv and the bench lines printed under the profiler.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/prof_other
rm -rf $out; mkdir -p $out
run() {  # tag, bench args...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python bench.py --no-cpu-baseline "$@" > $out/${tag}_bench.json 2> $out/$tag.err || { tail -3 $out/$tag.err; return 1; }
  find $out/$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_kernel_stats.csv
  python - "$out/${tag}_bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); n=d["steps"]
print(d["config"]["workload"], "|", round(d["ms_per_step"],3), "ms", {k: round(v/n,3) for k,v in d["kernel_ms"].items()})
PY
}
run m8 --steps 50 --warmup 5 --local-dims 32 32 32 32 --m 8 --shifts 1 &&
run m32 --steps 6 --warmup 2 --config 4 &&
run cap128 --steps 4 --warmup 2 --local-dims 64 64 64 128 --capacity 32
