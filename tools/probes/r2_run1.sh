cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "mlp_fused or layernorm_slabs or wgrad_group or gemm_glds or pair" > gpurun_out/r2a/ops.log 2>&1; echo "ops rc $?" 
tail -5 gpurun_out/r2a/ops.log
timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 50 > gpurun_out/r2a/bench_fused.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/r2a/bench_fused.log | cut -c1-400
VITAE_FUSE_MLP=0 timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 50 > gpurun_out/r2a/bench_unfused.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/r2a/bench_unfused.log | cut -c1-400
VITAE_WGRAD_SIDE=0 timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 50 > gpurun_out/r2a/bench_fused_noside.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/r2a/bench_fused_noside.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r2a/model.log 2>&1; echo "model rc $?"; tail -5 gpurun_out/r2a/model.log
