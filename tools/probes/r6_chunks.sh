cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
run() { echo "$1 b$2: $(env $1 python bench.py --batch $2 --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"; }
{
for c in 1 2 3 4 6 12; do run VITAE_ENC_CHUNKS=$c 4; done
for c in 1 2; do run VITAE_DEC_CHUNKS=$c 4; done
run VITAE_LN_FLUSH_ONCE=0 4
run VITAE_LN_FLUSH_ONCE=1 4
run VITAE_ADAMW_MAX_BLOCKS=512 4
run VITAE_ADAMW_MAX_BLOCKS=1024 4
run VITAE_ADAMW_MAX_BLOCKS=2048 4
} 2>&1 | tee gpurun_out/r6/chunks.txt
