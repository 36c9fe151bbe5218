# uneven encoder-backward buckets: a smaller last (exposed) bucket
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --profile-steps 0 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for e in "X=1" "VITAE_ENC_CUTS=0,2,7,12" "VITAE_ENC_CUTS=0,1,6,12" "VITAE_ENC_CUTS=0,3,7,12" "VITAE_ENC_CHUNKS=4 VITAE_ENC_CUTS=0,1,4,8,12" "VITAE_ENC_CHUNKS=4 VITAE_ENC_CUTS=0,2,5,8,12" "VITAE_ENC_CHUNKS=2 VITAE_ENC_CUTS=0,2,12" "X=2"; do
  a=$(env $e bash -c "$(declare -f run); run"); b=$(env $e bash -c "$(declare -f run); run"); echo "$e  $a $b"
done
