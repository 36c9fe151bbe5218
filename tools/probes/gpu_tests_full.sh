# the whole GPU suite with its own exit status and summary line kept (gpurun_out/gputests.txt)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/gputests.txt 2>&1; rc=$?
grep -E "passed|failed|error" gpurun_out/gputests.txt | tail -3
exit $rc
