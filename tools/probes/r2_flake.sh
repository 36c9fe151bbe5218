cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
for i in 1 2 3 4 5 6 7 8; do
  timeout 300 python -m pytest tests/test_augment.py tests/test_gpu_model.py -x -q -m gpu -k "augment or data_parallel_machinery or three_steps or config1" > gpurun_out/r2g/flake_$i.log 2>&1
  echo "run $i rc $? $(grep -E 'passed|failed|Fatal' gpurun_out/r2g/flake_$i.log | tail -1)"
done
