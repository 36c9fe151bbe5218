# Round-3 profile collection: default bench line, rocprofv3 --kernel-trace --stats of the same command (short form), timeline of one
# replayed step, PMC passes (FETCH_SIZE, WRITE_SIZE, SQ: separate runs, no trace domains beside them) at batch 4 and batch 32,
# the patch-8 step's kernel table, and the calibration of the GEMM families against the vendor library.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final3; mkdir -p $O
( time python bench.py ) > $O/bench_default.log 2>&1
grep '^{' $O/bench_default.log | tail -1 > $O/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5 > $O/stats.log 2>&1
python tools/timeline.py $(ls $O/stats/*/*kernel_trace.csv | head -1) 25 > $O/timeline.txt 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/stats
for B in 4 32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --batch $B --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > $O/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --batch $B --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > $O/pmc_write.log 2>&1
  python tools/summarize_pmc.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic_b$B.txt 2>&1
  rm -rf $O/pmc_fetch $O/pmc_write
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -- python bench.py --batch 32 --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > $O/pmc_sq.log 2>&1
python tools/summarize_pmc.py $O/pmc_sq > $O/pmc_sq_b32.txt 2>&1
rm -rf $O/pmc_sq
for cfg in "32:16:b32" "4:8:p8"; do
  B=${cfg%%:*}; r=${cfg#*:}; P=${r%%:*}; tag=${r#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$tag -- python bench.py --batch $B --patch $P --steps 10 --warmup 4 --no-extra --no-cpu-baseline --profile-steps 0 > $O/st_$tag.log 2>&1
  python tools/prof_summary.py $O/st_$tag 60 > $O/kernels_$tag.txt 2>&1
  rm -rf $O/st_$tag
done
mkdir -p $O/libtrace
PYTHONPATH=. rocprofv3 --kernel-trace --output-format csv -d $O/libtrace/tr -- python tools/gemm_vs_library.py > $O/libtrace/run.log 2>&1
python tools/libtrace_table.py $(ls $O/libtrace/tr/*/*kernel_trace.csv | head -1) > $O/gemm_vs_library.txt 2>&1
rm -rf $O/libtrace/tr
grep real $O/bench_default.log; cut -c1-300 $O/bench_line.json
