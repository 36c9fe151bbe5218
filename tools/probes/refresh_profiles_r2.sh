# Round-2 profile collection: the default bench line, rocprofv3 --kernel-trace --stats of the same command (short form),
# a kernel-by-kernel timeline of one replayed step, and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace
# domains beside them).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final2; mkdir -p $O
( time python bench.py ) > $O/bench_default.log 2>&1
grep '^{' $O/bench_default.log | tail -1 > $O/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5 > $O/stats.log 2>&1
python tools/timeline.py $(ls $O/stats/*/*kernel_trace.csv | head -1) 25 > $O/timeline.txt 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > $O/pmc_write.log 2>&1
python tools/summarize_pmc.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -- python bench.py --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > $O/pmc_sq.log 2>&1
python tools/summarize_pmc.py $O/pmc_sq > $O/pmc_sq.txt 2>&1
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
grep real $O/bench_default.log; cut -c1-300 $O/bench_line.json; head -12 $O/kernel_stats.csv | cut -c1-160
