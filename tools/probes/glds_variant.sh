#!/bin/bash
# Rebuilds libvitae_hip.so with extra -D flags for gemm_glds.hip and runs the GEMM microbench (cold weights).
# usage: tools/probes/glds_variant.sh "-DVITAE_GLDS_NS=3" [bench args...]
set -e
cd "$(dirname "$0")/../.."
PKG=vit_ae_plus_plus_amd
FLAGS="$1"; shift
cp $PKG/libvitae_hip.so /tmp/libvitae_full.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $FLAGS -I include -I $PKG/csrc -c $PKG/csrc/gemm_glds.hip -o /tmp/glds_var.o
objs=""
for f in gemm gemm_bf16 norm attention attention_mfma tokens loss optim; do objs="$objs $PKG/csrc/_obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libvitae_hip.so /tmp/glds_var.o $objs
echo "=== variant $FLAGS"
for a in "$@"; do python tools/gemm_bench.py $a --cold 2>&1 | grep -v amdgpu.ids; done
cp /tmp/libvitae_full.so $PKG/libvitae_hip.so
