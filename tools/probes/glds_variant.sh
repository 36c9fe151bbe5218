#!/bin/bash
# Rebuilds libvitae_hip.so with extra -D flags for gemm_glds.hip, runs the given commands, restores the library.
# usage: tools/probes/glds_variant.sh "-DVITAE_GLDS_NS=2" "python tools/gemm_bench.py pair" ["python tools/gemm_bench.py glds --cold" ...]
set -e
cd "$(dirname "$0")/../.."
PKG=vit_ae_plus_plus_amd
FLAGS="$1"; shift
cp $PKG/libvitae_hip.so /tmp/libvitae_full.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $FLAGS -I include -I $PKG/csrc -c $PKG/csrc/gemm_glds.hip -o /tmp/glds_var.o
objs=""
for f in $PKG/csrc/_obj/*.o; do case $f in */gemm_glds.o) ;; *) objs="$objs $f";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libvitae_hip.so /tmp/glds_var.o $objs
echo "=== variant $FLAGS"
for a in "$@"; do $a 2>&1 | grep -v amdgpu.ids; done
cp /tmp/libvitae_full.so $PKG/libvitae_hip.so
