# row-split BatchNorm and the gather kernel with four loads in flight: tests, kernel times, step A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/bng; O=gpurun_out/bng
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bn1d or gather" 2>&1 | tail -2
python tools/bn_bench.py 2>&1 | grep -v amdgpu | tee $O/bn.txt
for r in 1 2; do for v in "100000 build/variants/lib_tokold.so" "1024 default"; do set -- $v; L=$2; [ "$L" = default ] && L=""
  for cfg in 32:16 4:8 4:16; do B=${cfg%%:*}; P=${cfg#*:}
  VITAE_BN_SPLIT_MIN_ROWS=$1 VITAE_HIP_LIB=$L python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps 30 --warmup 8 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bn_split_min=$1 lib=$2 B=$B P=$P', d['ms_per_step'], 'ms')"
done; done; done | tee $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "config1 or micro or b4_fused or patch8" 2>&1 | tail -2
