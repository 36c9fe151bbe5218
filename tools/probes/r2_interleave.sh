# A/B of VITAE_GLDS_INTERLEAVE (DMA pieces between the MFMAs of a k-step) on the step time; rebuilds gemm_glds.hip on the box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
run() { python bench.py --no-cpu-baseline --no-extra --steps 100 --profile-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "interleave=1: $(run) $(run)"
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_glds or pair or wgrad_group or linear" 2>&1 | tail -1
touch vit_ae_plus_plus_amd/csrc/gemm_glds.hip
VITAE_HIPCC_FLAGS="-DVITAE_GLDS_INTERLEAVE=0" python -m vit_ae_plus_plus_amd.build > /dev/null 2>&1
echo "interleave=0: $(run) $(run)"
