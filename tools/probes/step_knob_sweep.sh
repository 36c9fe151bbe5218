run() { python bench.py --no-cpu-baseline --no-extra --steps 80 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for e in "X=1" "VITAE_PAIR_SPLIT_TARGET=6" "VITAE_PAIR_SPLIT_TARGET=8" "VITAE_PAIR_SPLIT_TARGET=14" "VITAE_PAIR_SPLIT_TARGET=0" "VITAE_GLDS_SPLIT_BLOCKS=256" "VITAE_GLDS_SPLIT_BLOCKS=512" "VITAE_GLDS_SPLIT_BLOCKS=1" "VITAE_GLDS_SPLIT_MIN_KT=4" "VITAE_GLDS_SPLIT_MIN_KT=12" "X=2"; do
  a=$(env $e bash -c "$(declare -f run); run"); b=$(env $e bash -c "$(declare -f run); run"); echo "$e  $a $b"
done
