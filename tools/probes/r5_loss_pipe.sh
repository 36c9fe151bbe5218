# loss_fwd_bwd with the z-exchange software-pipelined by one row against the previous build (build/variants/lib_lossold.so)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lossx; O=gpurun_out/lossx
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "loss or sobel" 2>&1 | tail -3
for r in 1 2; do
for cfg in 4:16 32:16 4:8; do
  B=${cfg%%:*}; P=${cfg#*:}
  for L in build/variants/lib_lossold.so ""; do
    echo "== B=$B p=$P lib=${L:-new}"
    env VITAE_HIP_LIB=$L LB_BATCH=$B LB_PATCH=$P LB_ONLY="one pass,gradient only" python tools/loss_bench.py 2>&1 | grep -v amdgpu
  done
done
done | tee $O/times_pipe.txt
