cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 60 --profile-steps 0 $EXTRA 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
export VITAE_FORCE_DDP=1 VITAE_DDP_SIM_BUSBW=300
for k in 0 1 2 3 4 5; do
  echo "skip $k: $(VITAE_DDP_SIM_SKIP=$k run)"
done
echo "skip 0 no-predictor-side: $(VITAE_PREDICTOR_SIDE=0 run)"
echo "skip 0 no-opt-in-backward: $(VITAE_OPT_IN_BACKWARD=0 run)"
