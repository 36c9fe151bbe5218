#!/bin/bash
# Rebuilds libvitae_hip.so with extra -D flags for gemm_glds.hip and runs the large-shape sweep (tools/gemm_big.py).
# usage: tools/probes/glds_big.sh "-DVITAE_GLDS_NS_T128=2" "VITAE_GLDS_T128=1" [forms...]
set -e
cd "$(dirname "$0")/../.."
PKG=vit_ae_plus_plus_amd
FLAGS="$1"; ENVS="$2"; shift; shift
cp $PKG/libvitae_hip.so /tmp/libvitae_full.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $FLAGS -I include -I $PKG/csrc -c $PKG/csrc/gemm_glds.hip -o /tmp/glds_var.o
objs=""
for f in $PKG/csrc/_obj/*.o; do case $f in */gemm_glds.o) ;; *) objs="$objs $f";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libvitae_hip.so /tmp/glds_var.o $objs
echo "=== variant $FLAGS  env $ENVS"
env $ENVS python tools/gemm_big.py "$@" 2>&1 | grep GLDS
cp /tmp/libvitae_full.so $PKG/libvitae_hip.so
