# the grouped weight-gradient launch on the wave-specialised 128 x 128 workgroup against the ping-pong one: tests, kind x split sweep, step A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wgroup; O=gpurun_out/wgroup
for k in 0 1; do VITAE_WGRAD_GROUP_WS=$k timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad_group" 2>&1 | tail -1; done
for k in 0 1; do for s in 1 2 3 4; do
  echo "== kind=$k split=$s"; VITAE_WGRAD_GROUP_WS=$k VITAE_WGRAD_GROUP_SPLIT=$s WG_ONLY_GROUP=1 timeout 300 python tools/wgrad_group_bench.py 2>&1 | grep -v amdgpu
done; done | tee $O/sweep.txt
echo "== auto"; WG_ONLY_GROUP=1 python tools/wgrad_group_bench.py 2>&1 | grep -v amdgpu | tee $O/auto.txt
for r in 1 2; do for k in 0 -1; do for cfg in 32:16 4:8 8:16; do
  B=${cfg%%:*}; P=${cfg#*:}
  VITAE_WGRAD_GROUP_WS=$k python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps 30 --warmup 8 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WS=$k B=$B P=$P', d['ms_per_step'], 'ms')"
done; done; done | tee $O/ab.txt
