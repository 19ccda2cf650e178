#!/bin/bash
# GPU box: hardware counters of the stencil kernels, one rocprofv3 pass per counter group (tools/hop_only.py).
# usage: [PMC_PASSES=n] [BCG_LIB=...] tools/pmc_passes.sh <tag> ["extra args for hop_only.py"]
# Every group below fits the hardware counter slots.  (A TA_BUSY / TA_*_STALLED / TD_TC_STALL group did not: rocprofv3
# aborted in rocprofiler_create_counter_config, error 38, before any kernel ran -- it is split into single-counter passes.)
export TMPDIR=/tmp
tag=$1; shift
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  [ $i -gt ${PMC_PASSES:-99} ] && break
  rm -rf gpurun_out/pmc_${tag}_$i
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -- python tools/hop_only.py $@ > gpurun_out/pmc_${tag}_$i.out 2> gpurun_out/pmc_${tag}_$i.err || { echo "pass $i failed: $group"; tail -3 gpurun_out/pmc_${tag}_$i.err; continue; }
  python tools/pmc_summary.py gpurun_out/pmc_${tag}_$i | grep "k_hop4"
done <<'GROUPS'
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TA_BUSY_avr
TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum
TD_TC_STALL_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUSY_avr
TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum
GROUPS
