# Round-5 profile collection (everything under gpurun_out/final5; the summaries are copied to profiles/round5_* by hand afterwards):
# default bench line; rocprofv3 --kernel-trace --stats of the same command; timelines at batch 4 / 32 / patch 8; the in-step tax
# tables (the same launches replayed from the graph / eager / without side branches / alone) with their SQ + GRBM + TCC counter passes
# at batch 4 / 32 / patch 8 (tools/probes/r5_tax.sh); FETCH_SIZE / WRITE_SIZE passes at batch 4; the SQ pass at batch 8 (north-star
# sub-total); attention / LayerNorm / loss tables; calibration against the vendor library; every tile family with the step's real
# epilogues (tools/epi_tiles.py).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/final5; mkdir -p $O
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
for cfg in 32:16:b32 4:8:p8 4:16:b4; do
  CFG=$cfg bash tools/probes/r5_tax.sh > /dev/null 2>&1
  tag=${cfg##*:}
  cp gpurun_out/r5tax_$tag/in_step_tax.txt $O/in_step_tax_$tag.txt
  cp gpurun_out/r5tax_$tag/pmc_sq.txt $O/pmc_sq_$tag.txt
done
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$pass -- python bench.py --batch 4 --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > $O/pmc_$pass.log 2>&1
done
python tools/summarize_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic_b4.txt 2>&1
python tools/pmc_round_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE gpurun_out/r5tax_b4/pmc_clk_counters.csv.gz $O/kernel_stats.csv > $O/gemm_traffic.json 2> $O/gemm_traffic.err
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
env VITAE_WGRAD_GROUP_SIDE=0 VITAE_PREDICTOR_SIDE=0 VITAE_OPT_IN_BACKWARD=0 VITAE_WGRAD_SIDE=0 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_b8 -- python bench.py --batch 8 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --profile-steps 0 --no-graph > $O/pmc_b8.log 2>&1
python tools/pmc_sq_json.py $O/pmc_b8 > $O/pmc_sq_b8.json 2> $O/pmc_sq_b8.err
python tools/pmc_table.py $O/pmc_b8 > $O/pmc_sq_b8.txt 2>&1
rm -rf $O/pmc_b8
python tools/attn_bench.py > $O/attention.txt 2>&1
python tools/ln_bench.py > $O/layernorm.txt 2>&1
python tools/loss_bench.py > $O/loss_b4.txt 2>&1; LB_BATCH=32 python tools/loss_bench.py > $O/loss_b32.txt 2>&1
mkdir -p $O/libtrace
rocprofv3 --kernel-trace --output-format csv -d $O/libtrace/tr -- python tools/gemm_vs_library.py > $O/libtrace/run.log 2>&1
python tools/libtrace_table.py $(ls $O/libtrace/tr/*/*kernel_trace.csv | head -1) > $O/gemm_vs_library.txt 2>&1
rm -rf $O/libtrace/tr
for b in 4 8 32; do python tools/epi_tiles.py B=$b 2>&1 | grep -v amdgpu > $O/epi_tiles_b$b.txt; done
grep real $O/bench_default.log; cut -c1-400 $O/bench_line.json; cat $O/gemm_vs_library.txt
