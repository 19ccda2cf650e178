#!/bin/bash
# Tuning aid: compile only the probed stencil instantiations (-DBCG_PROBE) and print their register use.
# usage: tools/probe_regs.sh [extra hipcc flags]
cd "$(dirname "$0")/../blockcg_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DBCG_PROBE "$@" -Rpass-analysis=kernel-resource-usage -c kernels_stencil.hip -o /tmp/km_probe.o 2> /tmp/km_probe.log
grep -i " error" -A4 /tmp/km_probe.log | head -20
python ../../tools/kernel_regs.py /tmp/km_probe.log "hop4"
