cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
for u in 8 12 4; do echo "unroll $u"; VITAE_ADAMW_UNROLL=$u python tools/optim_bench.py 2>&1 | grep adamw; done
for b in 256 512; do echo "blocks $b"; VITAE_ADAMW_MAX_BLOCKS=$b python tools/optim_bench.py 2>&1 | grep adamw; done
timeout 600 python -m pytest tests -x -q -m gpu -k "grad_norm_fused" 2>&1 | tail -2
