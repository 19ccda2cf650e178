#!/bin/bash
# GPU box: executed instruction counts of the stencil kernels per class (SQ_INSTS_*), tools/hop_only.py under rocprofv3.
# usage: [BCG_LIB=...] tools/pmc_hop_insts.sh <tag>
export TMPDIR=/tmp
tag=${1:-insts}
i=0
for g in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES" \
         "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_${tag}_$i
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -- python tools/hop_only.py 3 > gpurun_out/pmc_${tag}_$i.out 2> gpurun_out/pmc_${tag}_$i.err || { echo "pass failed: $g"; tail -3 gpurun_out/pmc_${tag}_$i.err; continue; }
  python tools/pmc_summary.py gpurun_out/pmc_${tag}_$i | grep "k_hop4b"
done
