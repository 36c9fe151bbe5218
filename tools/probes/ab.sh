# Alternating A/B of one environment knob on the step time (process-to-process spread is ~1.5 %: single runs prove nothing):
#   KNOB=VITAE_X VALS="0 1" ROUNDS=3 CFGS="4:8 32:16" bash tools/probes/ab.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CFGS=${CFGS:-4:8 32:16}
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in $VALS; do
    for cfg in $CFGS; do
      B=${cfg%%:*}; P=${cfg#*:}
      env $KNOB=$v python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps ${STEPS:-40} --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$KNOB=$v B=$B P=$P', d['ms_per_step'], 'ms')"
    done
  done
done | tee gpurun_out/ab_$KNOB.txt
python - <<PY
import collections,re
d=collections.defaultdict(list)
for l in open('gpurun_out/ab_$KNOB.txt'):
    m=re.match(r'(\S+) (B=\d+ P=\d+) ([\d.]+) ms',l)
    if m: d[(m.group(2),m.group(1))].append(float(m.group(3)))
for k,v in sorted(d.items()): print(k, 'min %.3f median %.3f'%(min(v), sorted(v)[len(v)//2]), v)
PY
