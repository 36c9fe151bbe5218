cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
bash tools/probes/gpu_tests_r6.sh
bash tools/probes/refresh_profiles_r6.sh > /dev/null 2>&1
F=gpurun_out/final6
# the committed JSONs the bench line quotes are this run's: put them in place, then the line itself
for n in gemm_traffic.json pmc_sq_b8.json; do [ -s $F/$n ] && cp $F/$n profiles/round6_$n; done
( time python bench.py ) > $F/bench_default.log 2>&1
grep '^{' $F/bench_default.log | tail -1 > $F/bench_line.json
grep real $F/bench_default.log; cut -c1-220 $F/bench_line.json
