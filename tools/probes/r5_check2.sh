# after the grouped ws weight gradients + loss kernel changes: loss kernel times, bench at batch 4 / 8 / 32 / patch 8, the whole GPU suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/chk2; O=gpurun_out/chk2
for cfg in 4:16 32:16; do B=${cfg%%:*}; P=${cfg#*:}; echo "== B=$B p=$P"; LB_BATCH=$B LB_PATCH=$P LB_ONLY="one pass,gradient only" python tools/loss_bench.py 2>&1 | grep -v amdgpu; done | tee $O/loss.txt
for cfg in 4:16 8:16 32:16 4:8; do B=${cfg%%:*}; P=${cfg#*:}
  python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps 40 --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B P=$P', d['ms_per_step'], 'ms', d['value'])"
done | tee $O/bench.txt
bash tools/probes/gpu_tests_full.sh
