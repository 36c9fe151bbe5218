# re-sweep of the GEMM knobs after the k-loops stopped draining their DMA queues (the optimum may have moved)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --profile-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for e in "X=1" "VITAE_GLDS_WIDE_MIN_TILES=250" "VITAE_GLDS_WIDE_MIN_TILES=150" "VITAE_PAIR_SPLIT_TARGET=6" "VITAE_PAIR_SPLIT_TARGET=14" "VITAE_PAIR_SPLIT_TARGET=0" "VITAE_GLDS_SPLIT_BLOCKS=256" "VITAE_GLDS_SPLIT_BLOCKS=600" "VITAE_GLDS_SPLIT_MIN_KT=4" "VITAE_ENC_CHUNKS=2" "VITAE_ENC_CHUNKS=4" "VITAE_OPT_IN_BACKWARD=0" "VITAE_PREDICTOR_SIDE=0" "X=2"; do
  a=$(env $e bash -c "$(declare -f run); run"); b=$(env $e bash -c "$(declare -f run); run"); echo "$e  $a $b"
done
