cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
run() { echo "$1 b$2: $(env $1 python bench.py --batch $2 --no-extra --no-cpu-baseline --steps 40 --warmup 10 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"; }
{
for rep in 1 2; do
run VITAE_SIDE_STREAMS=all 4
run VITAE_SIDE_STREAMS=none 4
run VITAE_SIDE_STREAMS=side 4
run VITAE_SIDE_STREAMS=oside 4
run VITAE_SIDE_STREAMS=pside 4
run VITAE_SIDE_STREAMS=auto 4
done
run VITAE_SIDE_STREAMS=all 32
run VITAE_SIDE_STREAMS=none 32
run VITAE_SIDE_STREAMS=wside 32
run VITAE_SIDE_STREAMS=wside,oside 32
run VITAE_SIDE_STREAMS=wside,side 32
run VITAE_SIDE_STREAMS=wside,pside 32
run VITAE_SIDE_STREAMS=auto 32
} 2>&1 | tee gpurun_out/r6/side_ab.txt
