cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r2l/tests.log 2>&1; echo "ops tests rc $? $(grep -E 'passed|failed' gpurun_out/r2l/tests.log | tail -1)"
grep -E "Error|assert |error" gpurun_out/r2l/tests.log | head -6
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --profile-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for e in "VITAE_GLDS_PIPE_MAX_WGS=512" "VITAE_GLDS_PIPE_MAX_WGS=0" "VITAE_GLDS_PIPE_MAX_WGS=300" "VITAE_GLDS_PIPE_MAX_WGS=1024" "VITAE_GLDS_PIPE_MAX_WGS=512" "VITAE_GLDS_PIPE_MAX_WGS=0"; do
  a=$(env $e bash -c "$(declare -f run); run"); b=$(env $e bash -c "$(declare -f run); run"); echo "$e  $a $b"
done
python tools/gemm_phase_probe.py 2>&1 | grep -E "^[a-z].*:" | sed 's/kernel span [0-9]* clk; start skew [0-9-]*; //' | cut -c1-330
