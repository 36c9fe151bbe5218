cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r2m/tests.log 2>&1; echo "ops tests rc $? $(grep -E 'passed|failed' gpurun_out/r2m/tests.log | tail -1)"
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --profile-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "step ms: $(run) $(run) $(run) $(run)"
python tools/gemm_phase_probe.py 2>&1 | grep -E "^[a-z].*:" | sed 's/kernel span [0-9]* clk; start skew [0-9-]*; //; s/step 4:.*mfma issue [0-9]*, //' | cut -c1-260
