cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --profile-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "step ms: $(run) $(run)"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -2
