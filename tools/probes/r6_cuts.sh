cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --batch ${B:-4} --no-cpu-baseline --no-extra --steps 60 --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'B=${B:-4}', d['ms_per_step'], 'ms')"; }
for r in 1 2; do
run X=1
run VITAE_ENC_CHUNKS=4 VITAE_ENC_CUTS=0,1,4,8,12
run VITAE_ENC_CHUNKS=4 VITAE_ENC_CUTS=0,2,5,8,12
run VITAE_ENC_CHUNKS=3 VITAE_ENC_CUTS=0,2,7,12
run VITAE_ENC_CHUNKS=3 VITAE_ENC_CUTS=0,1,6,12
run VITAE_ENC_CHUNKS=2
run VITAE_ENC_CHUNKS=4
run VITAE_ENC_CHUNKS=6
done
