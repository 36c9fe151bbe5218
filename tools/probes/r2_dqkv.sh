cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r2n/tests.log 2>&1; echo "tests rc $? $(grep -E 'passed|failed' gpurun_out/r2n/tests.log | tail -1)"
grep -E "^E  |Error" gpurun_out/r2n/tests.log | head -6
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --profile-steps 0 $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "B4: $(run) $(run) $(run)"
echo "B32: $(run '--batch 32 --steps 30')"
