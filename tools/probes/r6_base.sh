# round 6, first look: node overhead of a graph chain, the pair launches alone, the batch-4 step of the unchanged build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
O=gpurun_out/r6
./build/probes/chain_rate > $O/chain_rate.txt 2>&1
python tools/pair_bench.py B=4 > $O/pair_b4.txt 2>&1
python tools/pair_bench.py B=8 > $O/pair_b8.txt 2>&1
python tools/epi_tiles.py B=4 tiles=5,-1 > $O/epi_b4.txt 2>&1
for i in 1 2; do
python bench.py --batch 4 --no-cpu-baseline --no-extra --steps 60 --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=4', d['value'], 'vol/s', d['ms_per_step'], 'ms')"
done | tee $O/step_b4.txt
cat $O/chain_rate.txt $O/pair_b4.txt $O/pair_b8.txt $O/epi_b4.txt
