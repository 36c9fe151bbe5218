cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4attn; mkdir -p $O
for v in ${TESTV:-0 2}; do VITAE_ATTN_FWD=$v timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "sdpa" 2>&1 | tail -2; done
for v in ${BENCHV:-0 1 2}; do for rb in ${RBS:-1}; do echo "== VITAE_ATTN_FWD=$v RB=$rb"; VITAE_ATTN_RB=$rb VITAE_ATTN_FWD=$v timeout 300 python tools/attn_bench.py 2>/dev/null | grep "${SHAPES:-N=}"; done; done 2>&1 | tee $O/bench_fwd.txt
