#!/bin/bash
# GPU box: L2->fabric read requests per lattice site of the two stencil kernels for several lattice shapes
# (is the re-fetch rate a property of power-of-two strides?).  usage: tools/shape_fetch.sh "64 64 64 64" "48 56 56 64" ...
export TMPDIR=/tmp
n=0
for dims in "$@"; do
  n=$((n+1))
  rm -rf gpurun_out/shape_$n
  echo "== $dims: $(python tools/hop_only.py 4 $dims 2>/dev/null)"
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/shape_$n -- python tools/hop_only.py 4 $dims > /dev/null 2> gpurun_out/shape_$n.err
  python tools/pmc_summary.py gpurun_out/shape_$n | grep k_hop4 | python -c "
import sys
V=1
for d in '$dims'.split(): V*=int(d)
for l in sys.stdin:
    if 'TCC_EA0_RDREQ_sum' not in l: continue
    name=l.split('TCC_EA0_RDREQ_sum')[0].strip(); p=l.split()
    val=float(p[p.index('avg/launch')+1]); ms=float(p[-1].split('=')[1])
    print('   ', name, 'fetch B/site %.0f (= %.2f field reads + links 576)' % (val*128/V, (val*128/V-576)/768), 'ms', ms, 'ns/site %.3f' % (ms*1e6/V))
"
done
