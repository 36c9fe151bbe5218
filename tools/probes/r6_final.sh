cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
bash tools/probes/gpu_tests_r6.sh
