# XCD-aware piece map of the two one-pass loss kernels (VITAE_LOSS_XCD / VITAE_TARGET_XCD = 0: identity map) and the 4-slot load ring
# (build/variants/lib_lossd4.so): tests, kernel times at batch 4 / 32 / patch 8, FETCH_SIZE per launch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lossx; O=gpurun_out/lossx
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "loss or sobel" 2>&1 | tail -3
for r in 1 2; do
for cfg in 4:16 32:16 4:8; do
  B=${cfg%%:*}; P=${cfg#*:}
  for v in "0 default" "1 default" "1 build/variants/lib_lossd4.so"; do
    set -- $v; L=$2; [ "$L" = default ] && L=""
    echo "== B=$B p=$P xcd=$1 lib=$2"
    env VITAE_HIP_LIB=$L VITAE_LOSS_XCD=$1 VITAE_TARGET_XCD=$1 LB_BATCH=$B LB_PATCH=$P LB_ONLY="one pass,gradient only" python tools/loss_bench.py 2>&1 | grep -v amdgpu
  done
done
done | tee $O/times.txt
for x in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  env VITAE_LOSS_XCD=$x VITAE_TARGET_XCD=$x LB_BATCH=4 LB_ONLY="one pass,gradient only" rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$x -- python tools/loss_bench.py > /dev/null 2>&1
  echo "== xcd=$x $c"; python tools/summarize_pmc.py $O/pmc_${c}_$x 2>/dev/null | grep -i "loss_fwd_bwd\|target_edge"
  rm -rf $O/pmc_${c}_$x
done; done | tee $O/fetch.txt
