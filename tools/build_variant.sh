#!/bin/bash
# Build a variant of libvitae_hip.so with one source recompiled under extra flags (kernel experiments):
#   tools/build_variant.sh <name> <source.hip> <flags...>   ->  gpurun_out/variants/libvitae_<name>.so   (use with VITAE_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p build/variants
obj=build/variants/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I include -I vit_ae_plus_plus_amd/csrc "$@" -c vit_ae_plus_plus_amd/csrc/$src -o $obj
objs=$(ls vit_ae_plus_plus_amd/csrc/_obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libvitae_$name.so $objs $obj -ldl
echo build/variants/libvitae_$name.so
