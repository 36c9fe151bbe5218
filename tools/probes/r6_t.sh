cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -x -q -m gpu ${TESTK:+-k "$TESTK"} > gpurun_out/r6/test_out.txt 2>&1
echo rc=$?
grep -n "Error\|error\|assert\|FAILED\|passed\|failed" gpurun_out/r6/test_out.txt | head -40
