#!/bin/bash
# Builds ablated variants of gemm_bf16.hip (no global loads after the first phase / no MFMA loop) and times
# them with tools/gemm_bench.py to see which part of a phase sets the per-GEMM time.
set -e
cd "$(dirname "$0")/../.."
PKG=vit_ae_plus_plus_amd
cp $PKG/libvitae_hip.so /tmp/libvitae_full.so
for v in NOLOAD NOCOMPUTE; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVITAE_ABLATE_$v -I include -I $PKG/csrc -c $PKG/csrc/gemm_bf16.hip -o /tmp/gemm_$v.o
  objs=""
  for f in gemm norm attention attention_mfma tokens loss optim gemm_glds; do
    [ -f /tmp/o_$f.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $PKG/csrc -c $PKG/csrc/$f.hip -o /tmp/o_$f.o
    objs="$objs /tmp/o_$f.o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libvitae_hip.so /tmp/gemm_$v.o $objs
  echo "=== variant $v"
  python tools/gemm_bench.py 2>&1 | grep -E "enc.qkv|enc.proj|dec.fc1|pred " | grep -E "fwd|wgrad"
done
cp /tmp/libvitae_full.so $PKG/libvitae_hip.so
echo "=== full"
python tools/gemm_bench.py 2>&1 | grep -E "enc.qkv|enc.proj|dec.fc1|pred " | grep -E "fwd|wgrad"
