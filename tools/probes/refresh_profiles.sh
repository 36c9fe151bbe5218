cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( time python bench.py ) > gpurun_out/final/bench_default.log 2>&1
tail -1 gpurun_out/final/bench_default.log | grep '^{' > gpurun_out/final/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -- python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5 > gpurun_out/final/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_fetch -- python bench.py --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > gpurun_out/final/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_write -- python bench.py --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > gpurun_out/final/pmc_write.log 2>&1
python tools/prof_summary.py gpurun_out/final/stats 40
python tools/summarize_pmc.py gpurun_out/final/pmc_fetch gpurun_out/final/pmc_write | grep "==\|gemm_glds" 
grep real gpurun_out/final/bench_default.log
