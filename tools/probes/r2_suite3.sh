cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
for i in 1 2 3; do
  timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2k/suite_$i.log 2>&1
  echo "run $i rc $? $(grep -E 'passed|failed|Fatal' gpurun_out/r2k/suite_$i.log | tail -1)"
done
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
