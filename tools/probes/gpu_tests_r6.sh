# the whole GPU suite with its exit status (natural order, as the driver runs it); START=<test id> resumes behind a fixed failure
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -x -q -m gpu ${TESTK:+-k "$TESTK"} > gpurun_out/r6/gpu_tests_full.txt 2>&1
echo rc=$?
grep -n "Error\|FAILED\|passed\|failed" gpurun_out/r6/gpu_tests_full.txt | head -20; tail -3 gpurun_out/r6/gpu_tests_full.txt
