cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
O=gpurun_out/r6
python tools/pair_bench.py B=4 sq=1 2>/dev/null | tee $O/pair_b4_sq1.txt
python tools/pair_bench.py B=4 sq=0 2>/dev/null | tee $O/pair_b4_sq0.txt
for i in 1 2; do for v in 1 0; do
VITAE_EPI_GRADNORM=$v python bench.py --batch 4 --no-cpu-baseline --no-extra --steps 60 --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('EPI_GRADNORM=$v B=4', d['value'], 'vol/s', d['ms_per_step'], 'ms')"
done; done | tee $O/step_b4_sq.txt
