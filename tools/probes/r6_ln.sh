cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
O=gpurun_out/r6
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "micro_vs or config1 or bench_workload or gradient_accumulation or data_parallel or three_steps or grad_norm" 2>&1 | tail -3
for r in 1 2 3; do
for v in "0:1" "1.2e6:0"; do
for cfg in 4:16:60 8:16:40 32:16:20; do
  B=${cfg%%:*}; r2=${cfg#*:}; P=${r2%%:*}; S=${r2#*:}
  VITAE_LN_PART_MIN=${v%%:*} VITAE_LN_FLUSH_ONCE=${v#*:} python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps $S --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LN=$v B=$B P=$P', d['ms_per_step'], 'ms')"
done; done; done | tee $O/ln_cost.txt
python - <<PY
import collections,re
d=collections.defaultdict(list)
for l in open('gpurun_out/r6/ln_cost.txt'):
    m=re.match(r'LN=(\S+) (B=\d+ P=\d+) ([\d.]+) ms',l)
    if m: d[(m.group(2),m.group(1))].append(float(m.group(3)))
for k,v in sorted(d.items()): print(k, 'min %.3f median %.3f'%(min(v), sorted(v)[len(v)//2]), v)
PY
