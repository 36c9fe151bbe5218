cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "grad_norm_fused or three_steps" > gpurun_out/r5_newtests.txt 2>&1; echo "newtests rc $?"; tail -5 gpurun_out/r5_newtests.txt
# sustained-load clocks: rocm-smi sampled beside a long replayed batch-32 run
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/r5_smi_b32.txt 2>&1 &
SMI=$!
python bench.py --batch 32 --steps 1500 --warmup 20 --no-extra --no-cpu-baseline --profile-steps 0 > gpurun_out/r5_b32_long.log 2>&1
kill $SMI 2>/dev/null
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_b32_long.log | tail -1
CFG=32:16:b32 bash tools/probes/r5_tax.sh
CFG=4:8:p8 bash tools/probes/r5_tax.sh
CFG=4:16:b4 bash tools/probes/r5_tax.sh
