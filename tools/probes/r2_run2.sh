cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 1700 python -m pytest tests -x -q -m gpu -s > gpurun_out/r2d/gpu_tests.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|error|worst grad" gpurun_out/r2d/gpu_tests.log | tail -8
timeout 900 python bench.py > gpurun_out/r2d/bench_default.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/r2d/bench_default.log > gpurun_out/r2d/bench_line.json; cut -c1-600 gpurun_out/r2d/bench_line.json
timeout 120 python bench.py --gpus 2 --steps 3 > gpurun_out/r2d/bench_gpus2.log 2>&1; echo "gpus2 rc $? (expected non-zero on a 1-GPU box)"; tail -2 gpurun_out/r2d/bench_gpus2.log
