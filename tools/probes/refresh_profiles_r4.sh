# Round-4 profile collection: default bench line, rocprofv3 --kernel-trace --stats of the same command (short form), timeline of one
# replayed step at batch 4 / batch 32 / patch 8 with per-kernel tables, PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no trace
# domains beside them) at batch 4, attention / LayerNorm / loss kernel tables, calibration of the GEMM families against the vendor library.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final4; mkdir -p $O
( time python bench.py ) > $O/bench_default.log 2>&1
grep '^{' $O/bench_default.log | tail -1 > $O/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5 > $O/stats.log 2>&1
python tools/timeline.py $(ls $O/stats/*/*kernel_trace.csv | head -1) 30 > $O/timeline_b4.txt 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python tools/prof_summary.py $O/stats 8 > $O/kernels_b4.txt 2>&1
rm -rf $O/stats
for cfg in "32:16:b32" "4:8:p8"; do
  B=${cfg%%:*}; r=${cfg#*:}; P=${r%%:*}; tag=${r#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$tag -- python bench.py --batch $B --patch $P --steps 30 --warmup 5 --no-extra --no-cpu-baseline --profile-steps 0 > $O/st_$tag.log 2>&1
  python tools/timeline.py $(ls $O/st_$tag/*/*kernel_trace.csv | head -1) 30 > $O/timeline_$tag.txt 2>&1
  python tools/prof_summary.py $O/st_$tag 25 > $O/kernels_$tag.txt 2>&1
  rm -rf $O/st_$tag
done
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$pass -- python bench.py --batch 4 --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > $O/pmc_$pass.log 2>&1
done
python tools/summarize_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic_b4.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python tools/attn_bench.py > $O/attention.txt 2>&1
python tools/ln_bench.py > $O/layernorm.txt 2>&1
python tools/loss_bench.py > $O/loss_b4.txt 2>&1; LB_BATCH=32 python tools/loss_bench.py > $O/loss_b32.txt 2>&1
mkdir -p $O/libtrace
PYTHONPATH=. rocprofv3 --kernel-trace --output-format csv -d $O/libtrace/tr -- python tools/gemm_vs_library.py > $O/libtrace/run.log 2>&1
python tools/libtrace_table.py $(ls $O/libtrace/tr/*/*kernel_trace.csv | head -1) > $O/gemm_vs_library.txt 2>&1
rm -rf $O/libtrace/tr
grep real $O/bench_default.log; cut -c1-300 $O/bench_line.json
