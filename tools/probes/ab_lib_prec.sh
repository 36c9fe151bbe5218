# ab_lib.sh for another precision mode: PREC=fp32x3 LIBS="a.so default" ROUNDS=2 bash tools/probes/ab_lib_prec.sh
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${ROUNDS:-2}); do for v in $LIBS; do L=$v; [ "$v" = default ] && L=""
  env VITAE_HIP_LIB=$L python bench.py --precision ${PREC:-fp32x3} --no-cpu-baseline --no-extra --steps 40 --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LIB=$v', d['ms_per_step'], 'ms')"
done; done
