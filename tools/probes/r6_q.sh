cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
O=gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pair or sqnorm" 2>&1 | tail -3
for q in 1 0 1 0; do echo "WS64Q=$q"; VITAE_WS64Q=$q timeout 300 python tools/pair_bench.py B=4 2>/dev/null; done | tee $O/pair_q_b4.txt
for q in 1 0; do echo "WS64Q=$q"; VITAE_WS64Q=$q timeout 300 python tools/pair_bench.py B=8 2>/dev/null; done | tee $O/pair_q_b8.txt
for q in 0 1; do echo "==== VITAE_WS64Q=$q"; VITAE_WS64Q=$q python tools/ws64_phase_probe.py B=4 2>&1 | grep -v amdgpu.ids | grep -A3 "enc proj\|enc fc2" | grep -v "dgrad:\|wgrad:\|starts:" ; done | tee gpurun_out/r6/ws64_phase_q.txt
