# GEMM selection knobs at batch 8 (the encoder attention + MLP point of north_star) and 32
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 40 --profile-steps 0 --batch $B 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for B in 8 32; do
export B
for e in "X=1" "VITAE_GLDS_WIDE_MIN_TILES=150" "VITAE_GLDS_WIDE_MIN_TILES=250" "VITAE_GLDS_WIDE_MIN_TILES=330" "VITAE_GLDS_PIPE_MAX_WGS=768" "VITAE_GLDS_PIPE_MAX_WGS=1024" "VITAE_GLDS_PIPE_MAX_WGS=256" "VITAE_PAIR_SPLIT_TARGET=14" "VITAE_PAIR_SPLIT_TARGET=6" "VITAE_GLDS_SPLIT_BLOCKS=512" "VITAE_GLDS_SPLIT_BLOCKS=256" "X=2"; do
  a=$(env $e bash -c "$(declare -f run); run"); echo "B=$B $e  $a"
done
done
