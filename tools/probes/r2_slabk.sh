cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu -k "slab" > gpurun_out/r2h/tests.log 2>&1; echo "tests rc $? $(grep -E 'passed|failed' gpurun_out/r2h/tests.log | tail -1)"
grep -E "Error|assert|error" gpurun_out/r2h/tests.log | head -8
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --profile-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for e in "VITAE_SLAB_SPLITK=1" "VITAE_SLAB_SPLITK=0" "VITAE_SLABK_TARGET=256" "VITAE_SLABK_TARGET=512" "VITAE_SLABK_MIN_KT=2" "VITAE_SLABK_MIN_KT=6"; do
  a=$(env $e bash -c "$(declare -f run); run"); b=$(env $e bash -c "$(declare -f run); run"); echo "$e  $a $b"
done
