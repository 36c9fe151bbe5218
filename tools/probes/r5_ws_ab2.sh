cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_bt_forms or gemm_bt_split" 2>&1 | tail -2
for v in nointer default s3t0 s4t0 s5t0; do
  [ $v = default ] && unset VITAE_HIP_LIB || export VITAE_HIP_LIB=build/variants/lib_$v.so
  echo "== $v"
  for shp in "3520 768 3072 fwd" "3520 768 3072 dgrad" "3520 768 2304 dgrad" "6944 512 2048 fwd" "6944 512 2048 dgrad" "6944 512 512 fwd" "3520 768 768 fwd"; do
    python tools/ws_phase_probe.py $shp 2>/dev/null | grep -E "us/launch|B_9" | sed 's/; clocks.*//; s/consumer arrives.*producer;//' | tr '\n' ' '; echo
  done
done
