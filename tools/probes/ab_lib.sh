# Alternating A/B of variant LIBRARIES on the step time:  LIBS="default build/variants/lib_x.so" ROUNDS=3 CFGS="4:16" bash tools/probes/ab_lib.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in $LIBS; do
    for cfg in ${CFGS:-4:16}; do
      B=${cfg%%:*}; P=${cfg#*:}
      L=$v; [ "$v" = default ] && L=""
      env VITAE_HIP_LIB=$L python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps ${STEPS:-40} --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LIB=$v B=$B P=$P', d['ms_per_step'], 'ms')"
    done
  done
done | tee gpurun_out/ab_lib.txt
python - <<PY
import collections,re
d=collections.defaultdict(list)
for l in open('gpurun_out/ab_lib.txt'):
    m=re.match(r'(\S+) (B=\d+ P=\d+) ([\d.]+) ms',l)
    if m: d[(m.group(2),m.group(1))].append(float(m.group(3)))
for k,v in sorted(d.items()): print(k, 'min %.3f median %.3f'%(min(v), sorted(v)[len(v)//2]), v)
PY
