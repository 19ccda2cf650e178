set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/prof_cap; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/cap128 -- python bench.py --no-cpu-baseline --steps 4 --warmup 2 --local-dims 64 64 64 128 --capacity 32 > $out/cap128_bench.json 2> $out/cap128.err || tail -3 $out/cap128.err
find $out/cap128 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/cap128_kernel_stats.csv
python tools/half_volume_time.py > gpurun_out/half_volume_time.txt 2>&1; tail -1 gpurun_out/half_volume_time.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 1500 gpurun_out/bench_default.json
