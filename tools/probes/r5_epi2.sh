cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "linear_fwd_bwd or gemm_bt_forms or wsx3 or saved_preactivation or bwd_pair" > gpurun_out/r5_optests.txt 2>&1; echo "optests rc $?"; tail -4 gpurun_out/r5_optests.txt
PYTHONPATH=. python tools/epi_ablate.py B=32 tile=0 2>&1 | grep -v amdgpu
PYTHONPATH=. python tools/epi_ablate.py B=4 tile=5 2>&1 | grep -v amdgpu
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r5_modeltests.txt 2>&1; echo "modeltests rc $?"; tail -4 gpurun_out/r5_modeltests.txt
for b in 4 32; do for i in 1 2; do
VITAE_AUX_DERIV=0 VITAE_FC1_BIAS_BY_WGRAD=0 python bench.py --batch $b --steps 60 --warmup 10 --no-extra --no-cpu-baseline --profile-steps 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/b$b old /"
python bench.py --batch $b --steps 60 --warmup 10 --no-extra --no-cpu-baseline --profile-steps 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/b$b new /"
done; done
