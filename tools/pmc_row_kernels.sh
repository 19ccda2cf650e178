#!/bin/bash
# GPU box: SQ counters of the fused row kernels (phase B, phase C, k_phaseC_multi) over two full groups of bench.py:
# how busy the matrix pipe is and what the waves wait for.  usage: tools/pmc_row_kernels.sh <tag>
export TMPDIR=/tmp
tag=${1:-rows}
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_CYCLES" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_${tag}_$i
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -- python bench.py --steps 4 --warmup 4 --no-cpu-baseline > gpurun_out/pmc_${tag}_$i.out 2> gpurun_out/pmc_${tag}_$i.err || { echo "pass failed: $g"; tail -3 gpurun_out/pmc_${tag}_$i.err; continue; }
  python tools/pmc_summary.py gpurun_out/pmc_${tag}_$i | grep "k_phase"
done
