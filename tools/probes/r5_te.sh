cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "loss or sobel or piece_map" 2>&1 | tail -2
for r in 1 2; do for cfg in 4:16 32:16; do B=${cfg%%:*}; P=${cfg#*:}; echo "== B=$B p=$P"; LB_BATCH=$B LB_PATCH=$P LB_ONLY="target_edge" python tools/loss_bench.py 2>&1 | grep -v amdgpu; done; done
