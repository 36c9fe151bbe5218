cd $GRAFT_REPO_ROOT
for sel in "tests/test_gpu_model.py" "tests/test_ddp_streams.py tests/test_gpu_model.py" "tests/test_ddp_two_ranks_one_gpu.py tests/test_gpu_model.py" "tests/test_bench_plumbing.py tests/test_gpu_model.py" "tests/test_abi.py tests/test_gpu_model.py"; do
echo "== $sel"; python -m pytest $sel -m gpu -x -q -k "ddp or abi or plumbing or recurring" 2>&1 | grep -E "passed|failed|AssertionError: \(" | head -3
done
