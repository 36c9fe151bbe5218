cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
O=gpurun_out/r6
for r in 1 2 3; do
for w in "dec.fc1" "decoder" "decoder,enc.proj"; do
for cfg in 4:16:60 32:16:20 4:8:20 8:16:40; do
  B=${cfg%%:*}; r2=${cfg#*:}; P=${r2%%:*}; S=${r2#*:}
  VITAE_W2="$w" python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps $S --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('W2=$w B=$B P=$P', d['ms_per_step'], 'ms')"
done; done; done | tee $O/w2_cost.txt
python - <<PY
import collections,re
d=collections.defaultdict(list)
for l in open('gpurun_out/r6/w2_cost.txt'):
    m=re.match(r'W2=(\S+) (B=\d+ P=\d+) ([\d.]+) ms',l)
    if m: d[(m.group(2),m.group(1))].append(float(m.group(3)))
for k,v in sorted(d.items()): print(k, 'min %.3f median %.3f'%(min(v), sorted(v)[len(v)//2]), v)
PY
