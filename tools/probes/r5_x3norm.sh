cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "grad_norm or b4_fused or three_steps or accumulation" 2>&1 | tail -3
for r in 1 2; do for e in 0 1; do
  VITAE_EPI_GRADNORM=$e python bench.py --precision fp32x3 --no-cpu-baseline --no-extra --steps 30 --warmup 8 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32x3 epi_gradnorm=$e', d['ms_per_step'], 'ms', d['value'])"
done; done
