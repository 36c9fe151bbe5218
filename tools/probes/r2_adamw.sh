cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --profile-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for e in "X=1" "VITAE_ADAMW_MAX_BLOCKS=64" "VITAE_ADAMW_MAX_BLOCKS=128" "VITAE_ADAMW_MAX_BLOCKS=512" "VITAE_ADAMW_MAX_BLOCKS=64 VITAE_GRADNORM_MAX_BLOCKS=128" "VITAE_ADAMW_MAX_BLOCKS=32 VITAE_GRADNORM_MAX_BLOCKS=64 VITAE_ENC_CHUNKS=6" "VITAE_ADAMW_MAX_BLOCKS=64 VITAE_ENC_CHUNKS=6" "VITAE_ENC_CHUNKS=6"; do
  a=$(env $e bash -c "$(declare -f run); run"); b=$(env $e bash -c "$(declare -f run); run"); echo "$e  $a $b"
done
