#!/bin/bash
# GPU box: LDS-side counters of the two stencil launches with the links broadcast by DPP (default build) and read per lane
# (-DBCG_HOP4B_BCAST=0, tools/build_variant.sh nobcast), tools/hop_only.py, separate rocprofv3 --pmc passes per group.
export TMPDIR=/tmp
out=$PWD/gpurun_out/r04/bcast_pmc; rm -rf $out; mkdir -p $out
R=$PWD
cd /tmp
for v in base nobcast; do
  if [ "$v" = base ]; then unset BCG_LIB; else export BCG_LIB=$R/blockcg_amd/_build/libblockcg_hip_$v.so; fi
  n=0
  for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    n=$((n+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $out/${v}_$n -o pmc --output-format csv -- python3 $R/tools/hop_only.py 4 > $out/${v}_$n.log 2>&1 || { tail -5 $out/${v}_$n.log; exit 1; }
  done
done
cd $R
python3 tools/pmc_by_kernel.py $out | tee $out/summary.txt
