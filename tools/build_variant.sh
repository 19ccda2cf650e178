#!/bin/bash
# Build a variant of the library with extra compile flags: tools/build_variant.sh <name> "<flags>"
# -> blockcg_amd/_build/libblockcg_hip_<name>.so  (select with BCG_LIB or tools/ab_bench.sh)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../blockcg_amd/csrc"
mkdir -p /tmp/bcg_variants
# (the flags reach both kernel files: the stencil's switches live in kernels_stencil.hip, the row kernels' in kernels_mfma.hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c kernels_stencil.hip -o /tmp/bcg_variants/ks_$name.o &
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c kernels_mfma.hip -o /tmp/bcg_variants/km_$name.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../_build/libblockcg_hip_$name.so ../_build/capi_context.o ../_build/capi_operator.o ../_build/capi_solvers.o ../_build/kernels_generic.o /tmp/bcg_variants/km_$name.o /tmp/bcg_variants/ks_$name.o
echo built $name
