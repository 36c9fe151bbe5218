cd $GRAFT_REPO_ROOT
bash tools/probes/gpu_tests_r6.sh
run() { echo "$1 b$2: $(env $1 python bench.py --batch $2 --no-extra --no-cpu-baseline --steps 40 --warmup 10 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"; }
{ for i in 1 2; do run VITAE_WGRAD_GROUP_MIN=0.6e6 8; run VITAE_WGRAD_GROUP_MIN=1e9 8; run VITAE_WGRAD_GROUP_MIN=1.2e6 8; done; } > gpurun_out/r6/grp_b8.txt 2>&1
cat gpurun_out/r6/grp_b8.txt
