cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests -x -q -m gpu ${TESTK:+-k "$TESTK"} > gpurun_out/r6/test_out.txt 2>&1
echo rc=$?
head -50 gpurun_out/r6/test_out.txt; echo ...; tail -8 gpurun_out/r6/test_out.txt
