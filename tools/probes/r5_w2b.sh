cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "two_plane" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -s -k "bench_workload_b4 and bf16" 2>&1 | grep -E "loss errors|passed|failed|Error" | cut -c1-900
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do for cfg in "4 16" "8 16" "32 16" "4 8"; do set -- $cfg; for w in "" "dec.fc1"; do
VITAE_W2="$w" python bench.py --batch $1 --patch $2 --steps 50 --warmup 8 --no-extra --no-cpu-baseline --profile-steps 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/b$1 p$2 W2='$w' /"
done; done; done
