cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "cosine or bn1d" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -2
for cfg in 4:16 32:16; do B=${cfg%%:*}; P=${cfg#*:}
  python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps 40 --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B P=$P', d['ms_per_step'], 'ms', d['value'])"
done
