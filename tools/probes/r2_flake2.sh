cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
for i in 1 2 3 4; do
  timeout 600 python -m pytest tests/test_augment.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r2g/full_$i.log 2>&1
  echo "run $i rc $? $(grep -E 'passed|failed|Fatal' gpurun_out/r2g/full_$i.log | tail -1)"
done
