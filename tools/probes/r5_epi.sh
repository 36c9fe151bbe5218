cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "grad_norm_fused or three_steps" > gpurun_out/r5_newtests.txt 2>&1; echo "newtests rc $?"; tail -3 gpurun_out/r5_newtests.txt
PYTHONPATH=. python tools/epi_tiles.py B=32 > gpurun_out/r5_epi_tiles_b32.txt 2>&1
cat gpurun_out/r5_epi_tiles_b32.txt
PYTHONPATH=. python tools/epi_tiles.py B=8 > gpurun_out/r5_epi_tiles_b8.txt 2>&1
cat gpurun_out/r5_epi_tiles_b8.txt
