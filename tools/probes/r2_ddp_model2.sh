cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 60 --profile-steps 0 $EXTRA 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
export VITAE_FORCE_DDP=1
for q in 4 8 16; do
  echo "queues $q no model: $(GPU_MAX_HW_QUEUES=$q run)   plain (no ddp): $(VITAE_FORCE_DDP=0 GPU_MAX_HW_QUEUES=$q run)"
  for bw in 300 150; do
    a=$(env GPU_MAX_HW_QUEUES=$q VITAE_DDP_SIM_BUSBW=$bw bash -c "$(declare -f run); run"); echo "queues $q busbw $bw  $a"
  done
done
