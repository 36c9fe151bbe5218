cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "native_rccl or data_parallel_machinery or two_batch or fused_mlp_chain" > gpurun_out/r2e/ddp_tests.log 2>&1; echo "tests rc $?"
tail -15 gpurun_out/r2e/ddp_tests.log
VITAE_FORCE_DDP=1 timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 30 > gpurun_out/r2e/bench_forceddp.log 2>&1; echo "force ddp rc $?"; tail -1 gpurun_out/r2e/bench_forceddp.log | cut -c1-250
VITAE_FORCE_DDP=1 VITAE_DDP_NATIVE=1 timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 30 > gpurun_out/r2e/bench_forceddp_native.log 2>&1; echo "force ddp native rc $?"; tail -1 gpurun_out/r2e/bench_forceddp_native.log | cut -c1-250
