export TMPDIR=/tmp
for g in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); rm -rf gpurun_out/pmcx_$i
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcx_$i -- python tools/hop_only.py 3 > gpurun_out/pmcx_$i.out 2> gpurun_out/pmcx_$i.err || { echo "pass failed: $g"; tail -3 gpurun_out/pmcx_$i.err; continue; }
  python tools/pmc_summary.py gpurun_out/pmcx_$i | grep "k_hop4c<16, 0"
done
