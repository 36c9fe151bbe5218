#!/bin/bash
# Rebuilds libvitae_hip.so with extra -D flags for loss.hip and runs tools/loss_bench.py.
set -e
cd "$(dirname "$0")/../.."
PKG=vit_ae_plus_plus_amd
FLAGS="$1"
cp $PKG/libvitae_hip.so /tmp/libvitae_full.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $FLAGS -I include -I $PKG/csrc -c $PKG/csrc/loss.hip -o /tmp/loss_var.o
objs=""
for f in gemm gemm_bf16 gemm_glds norm attention attention_mfma tokens optim input; do objs="$objs $PKG/csrc/_obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libvitae_hip.so /tmp/loss_var.o $objs
echo "=== variant $FLAGS"
python tools/loss_bench.py 2>&1 | grep -v amdgpu.ids
cp /tmp/libvitae_full.so $PKG/libvitae_hip.so
