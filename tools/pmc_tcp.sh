#!/bin/bash
# GPU box: stencil time plus the L1 (TCP) stall counters.  usage: [env...] tools/pmc_tcp.sh <tag>
export TMPDIR=/tmp
tag=$1
echo "== $tag: $(python tools/hop_only.py 6 2>/dev/null)"
rm -rf gpurun_out/tcp_$tag
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/tcp_$tag -- python tools/hop_only.py 4 > /dev/null 2> gpurun_out/tcp_$tag.err
python tools/pmc_summary.py gpurun_out/tcp_$tag | grep "k_hop4" | awk '{printf "    "; for(i=1;i<=NF;i++) printf "%s ", $i; printf "\n"}'
