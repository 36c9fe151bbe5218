# quick step-level check: attention / pinned-model tests, then the batch-4, patch-8 and batch-32 points (no secondary data)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4step; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "${TESTK:-sdpa or patch8 or bench_workload or micro_vs}" 2>&1 | tail -3
for cfg in ${CFGS:-"4:16" "4:8" "32:16"}; do
  B=${cfg%%:*}; P=${cfg#*:}
  python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps 40 --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B P=$P', d['value'], 'vol/s', d['ms_per_step'], 'ms')"
done | tee -a $O/points.txt
