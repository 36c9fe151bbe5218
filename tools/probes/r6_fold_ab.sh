cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "cosine or adamw or gradnorm or opt_tail" > gpurun_out/r6/t_fold.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r6/t_fold.log | tail -5
python -m pytest tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r6/t_fold2.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r6/t_fold2.log | tail -5
run() { echo "$1 b$2: $(env $1 python bench.py --batch $2 --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"; }
for i in 1 2 3; do run VITAE_ADAMW_ACC_GATE=0 4; run VITAE_ADAMW_ACC_GATE=1 4; done
