#!/bin/bash
# One kernel file rebuilt with extra -D flags and linked with the other (already built) objects into build/variants/lib_<name>.so —
# for A/B timing of compile-time variants: VITAE_HIP_LIB=build/variants/lib_<name>.so python tools/...   (build/ travels with gpurun)
#   tools/variant_lib.sh <name> <source.hip> [-DFLAG=1 ...]
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p build/variants
extra=""
[ "$src" = attention_mfma.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
obj=build/variants/${name}_${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I include -I vit_ae_plus_plus_amd/csrc $extra "$@" -c vit_ae_plus_plus_amd/csrc/$src -o $obj
others=$(ls vit_ae_plus_plus_amd/csrc/_obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_${name}.so $others $obj -ldl
rm -f $obj
echo build/variants/lib_${name}.so
