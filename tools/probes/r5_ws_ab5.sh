cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=.
for v in noalias default s4 s5; do
  [ $v = default ] && unset VITAE_HIP_LIB || export VITAE_HIP_LIB=build/variants/lib_$v.so
  echo "== $v"
  timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_bt_forms and 4-" 2>&1 | tail -1
  for shp in "3520 768 3072 fwd" "3520 768 3072 dgrad" "3520 768 2304 dgrad" "6944 512 2048 fwd" "6944 512 2048 dgrad" "3520 768 768 fwd" "3072 768 3520 wgrad"; do
    python tools/ws_phase_probe.py $shp 2>/dev/null | grep -E "us/launch|whole loop" | sed 's/; clocks.*//' | tr '\n' ' '; echo
  done
done
