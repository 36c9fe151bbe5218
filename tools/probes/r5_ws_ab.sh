cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_bt_forms or gemm_bt_split" 2>&1 | tail -3
python tools/ws_phase_probe.py 3520 768 3072 | grep -v "B_[5-8]"
python tools/ws_phase_probe.py 3520 768 3072 dgrad | grep -v "B_[5-8]"
for v in old new; do
  [ $v = old ] && export VITAE_HIP_LIB=build/variants/lib_nointer.so || unset VITAE_HIP_LIB
  echo "== $v"
  python tools/bt_bench.py step forms=fwd,dgrad,wgrad tiles=4 2>/dev/null | grep -E "M= 3520|M= 6944|M=  768|M= 3072|M= 2304|M= 2048|M= 1536|M=  512" | grep -v "relerr.*!!!" | awk '{print $1,$2,$3,$4,$9,$10,$11,$12}' | head -60
done
