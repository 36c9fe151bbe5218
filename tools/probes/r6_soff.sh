cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
O=gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or pair or linear or two_plane or w2" 2>&1 | tail -3
for l in default build/variants/lib_soff0.so; do L=$l; [ "$l" = default ] && L=""; echo "LIB=$l"; VITAE_HIP_LIB=$L python tools/epi_tiles.py B=4 tiles=5 2>&1 | grep -v amdgpu | cut -c1-70; VITAE_HIP_LIB=$L python tools/pair_bench.py B=4 2>/dev/null | tail -1; done | tee $O/soff_tiles.txt
LIBS="default build/variants/lib_soff0.so" ROUNDS=3 CFGS="4:16 8:16 32:16" STEPS=40 bash tools/probes/ab_lib.sh 2>&1 | tail -8 | tee $O/soff_step.txt
