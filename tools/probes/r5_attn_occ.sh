# attention occupancy: 64-key chunk forward for N <= 64 and the one-launch backward cut to 168 VGPRs (three workgroups per CU):
# tests, kernel times (old build / min-waves 2 / new), step A/B at batch 32 / 8 / 4 and patch 8
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/attnocc; O=gpurun_out/attnocc
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "sdpa or attn or piece_map" 2>&1 | tail -2
for r in 1 2; do for L in build/variants/lib_attnold.so build/variants/lib_attnminw2.so ""; do echo "== lib=${L:-new}"; VITAE_HIP_LIB=$L python tools/attn_bench.py 2>&1 | grep -v amdgpu; done; done | tee $O/kernels.txt
for r in 1 2; do for L in build/variants/lib_attnold.so ""; do for cfg in 32:16 8:16 4:16 4:8; do B=${cfg%%:*}; P=${cfg#*:}
  VITAE_HIP_LIB=$L python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps 30 --warmup 8 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=${L:-new} B=$B P=$P', d['ms_per_step'], 'ms')"
done; done; done | tee $O/ab.txt
