#!/bin/bash
# Build a variant of the library with extra compile flags: tools/build_variant.sh <name> "<flags>"
# -> blockcg_amd/_build/libblockcg_hip_<name>.so  (select with BCG_LIB or tools/ab_bench.sh)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../blockcg_amd/csrc"
mkdir -p /tmp/bcg_variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c kernels_mfma.hip -o /tmp/bcg_variants/km_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../_build/libblockcg_hip_$name.so ../_build/blockcg_capi.o ../_build/kernels_generic.o /tmp/bcg_variants/km_$name.o
echo built $name
