cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=.
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_bt or bwd_pair or wgrad_group" 2>&1 | tail -2
for v in nointer default; do
  [ $v = default ] && unset VITAE_HIP_LIB || export VITAE_HIP_LIB=build/variants/lib_$v.so
  echo "== $v"
  for shp in "3520 768 3072 fwd" "3520 768 3072 dgrad" "3520 768 2304 dgrad" "6944 512 2048 fwd" "6944 512 2048 dgrad" "3520 768 768 fwd" "3072 768 3520 wgrad"; do
    python tools/ws_phase_probe.py $shp 2>/dev/null | grep -E "us/launch|whole loop" | sed 's/; clocks.*//' | tr '\n' ' '; echo
  done
done
unset VITAE_HIP_LIB
for cfg in "32 16" "4 8"; do set -- $cfg; for i in 1 2; do
VITAE_HIP_LIB=build/variants/lib_nointer.so python bench.py --batch $1 --patch $2 --steps 40 --warmup 8 --no-extra --no-cpu-baseline --profile-steps 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/b$1 p$2 nointer /"
python bench.py --batch $1 --patch $2 --steps 40 --warmup 8 --no-extra --no-cpu-baseline --profile-steps 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/b$1 p$2 new /"
done; done
