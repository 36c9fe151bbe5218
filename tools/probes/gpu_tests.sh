# usage: bash tools/probes/gpu_tests.sh [pytest args...]   -> summary lines; full log in gpurun_out/gpu_tests.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest "${@:-tests}" -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc $?"
grep -E "passed|failed|error|Error|assert " gpurun_out/gpu_tests.log | tail -12
