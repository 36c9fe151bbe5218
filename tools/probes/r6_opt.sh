cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
O=gpurun_out/r6
timeout 1500 python -m pytest tests -x -q -m gpu -k "adamw or gradnorm or micro_vs or config1 or bench_workload or gradient_accumulation or data_parallel or three_steps or grad_norm or checkpoint or two_plane" 2>&1 | tail -4
python tools/optim_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/optim_bench.txt
for r in 1 2 3; do
for v in 1 0; do
for cfg in 4:16:60 8:16:40 32:16:20; do
  B=${cfg%%:*}; r2=${cfg#*:}; P=${r2%%:*}; S=${r2#*:}
  VITAE_OPT_STATE16=$v python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps $S --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('STATE16=$v B=$B P=$P', d['ms_per_step'], 'ms')"
done; done; done | tee $O/state16_cost.txt
python - <<PY
import collections,re
d=collections.defaultdict(list)
for l in open('gpurun_out/r6/state16_cost.txt'):
    m=re.match(r'STATE16=(\S+) (B=\d+ P=\d+) ([\d.]+) ms',l)
    if m: d[(m.group(2),m.group(1))].append(float(m.group(3)))
for k,v in sorted(d.items()): print(k, 'min %.3f median %.3f'%(min(v), sorted(v)[len(v)//2]), v)
PY
