cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
O=gpurun_out/r6
timeout 900 python -m pytest tests -x -q -m gpu -k "sqnorm or adamw or gradnorm or grad_norm or pair or bench_workload or micro_vs or ws_128x256 or big_tile" 2>&1 | tail -5
for s in 2 1 0; do python tools/pair_bench.py B=4 sq=$s 2>/dev/null | tail -1; done | tee $O/pair_b4_spread.txt
python tools/pair_bench.py B=4 sq=2 2>/dev/null | tee -a $O/pair_b4_spread.txt
for i in 1 2 3; do for v in 1 0; do
VITAE_SQ_SPREAD=$v python bench.py --batch 4 --no-cpu-baseline --no-extra --steps 60 --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SQ_SPREAD=$v B=4', d['value'], 'vol/s', d['ms_per_step'], 'ms')"
done; done | tee $O/step_b4_spread.txt
for v in 1 0; do
VITAE_SQ_SPREAD=$v python bench.py --batch 8 --no-cpu-baseline --no-extra --steps 40 --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SQ_SPREAD=$v B=8', d['value'], 'vol/s', d['ms_per_step'], 'ms')"
VITAE_SQ_SPREAD=$v python bench.py --batch 32 --no-cpu-baseline --no-extra --steps 20 --warmup 5 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SQ_SPREAD=$v B=32', d['value'], 'vol/s', d['ms_per_step'], 'ms')"
done | tee -a $O/step_b4_spread.txt
