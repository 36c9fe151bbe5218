cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
python tools/ws64_phase_probe.py B=4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/ws64_phase_b4.txt
