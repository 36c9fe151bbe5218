cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
for q in 0 1; do echo "==== VITAE_WS64Q=$q"; VITAE_WS64Q=$q python tools/ws64_phase_probe.py B=4 2>&1 | grep -v amdgpu.ids | grep -A3 "enc proj\|enc fc2" | grep -v "dgrad:\|wgrad:\|starts:" ; done | tee gpurun_out/r6/ws64_phase_q.txt
