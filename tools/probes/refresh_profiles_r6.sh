# Round-6 profile collection (everything under gpurun_out/final6; the summaries are copied to profiles/round6_* afterwards):
# default bench line; rocprofv3 --kernel-trace --stats of the same command; timelines at batch 4 / 32 / patch 8; the in-step tax table
# with its SQ counter pass at batch 4; FETCH_SIZE / WRITE_SIZE passes at batch 4 (whole step, and the pair launches alone by shape);
# the SQ pass at batch 8 (north-star sub-total); pair launches alone (one tile per workgroup vs the persistent form, gradient-norm
# share on one address / spread slots / none); phase stamps of the pair launches; every tile family with the step's real epilogues;
# the graph-node overhead probe; AdamW pass.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/final6; mkdir -p $O
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
CFG=4:16:b4 bash tools/probes/r5_tax.sh > /dev/null 2>&1
cp gpurun_out/r5tax_b4/in_step_tax.txt $O/in_step_tax_b4.txt
cp gpurun_out/r5tax_b4/pmc_sq.txt $O/pmc_sq_b4.txt
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$pass -- python bench.py --batch 4 --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph > $O/pmc_$pass.log 2>&1
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pair_$pass -- python tools/pair_bench.py B=4 iters=2 > $O/pair_$pass.log 2>&1
done
python tools/summarize_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic_b4.txt 2>&1
python tools/pmc_round_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE gpurun_out/r5tax_b4/pmc_clk_counters.csv.gz $O/kernel_stats.csv > $O/gemm_traffic.json 2> $O/gemm_traffic.err
( echo "# FETCH_SIZE (KB, 64-byte units: x2 on gfx950) and WRITE_SIZE (KB) per launch of the pair kernels of tools/pair_bench.py B=4, by grid size"; python tools/pmc_by_grid.py $O/pair_FETCH_SIZE pair; python tools/pmc_by_grid.py $O/pair_WRITE_SIZE pair ) > $O/pair_traffic_by_shape.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pair_FETCH_SIZE $O/pair_WRITE_SIZE
env VITAE_WGRAD_GROUP_SIDE=0 VITAE_PREDICTOR_SIDE=0 VITAE_OPT_IN_BACKWARD=0 VITAE_WGRAD_SIDE=0 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_b8 -- python bench.py --batch 8 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --profile-steps 0 --no-graph > $O/pmc_b8.log 2>&1
python tools/pmc_sq_json.py $O/pmc_b8 > $O/pmc_sq_b8.json 2> $O/pmc_sq_b8.err
python tools/pmc_table.py $O/pmc_b8 > $O/pmc_sq_b8.txt 2>&1
rm -rf $O/pmc_b8
( for q in 0 1; do echo "=== VITAE_WS64Q=$q (0: one tile per workgroup, the default; 1: persistent workgroups)"; VITAE_WS64Q=$q python tools/pair_bench.py B=4 2>/dev/null; done
  for s in 1 0; do echo "=== gradient-norm share: sq=$s (1: one address, the form up to round 5; 0: none); spread slots above"; python tools/pair_bench.py B=4 sq=$s 2>/dev/null | tail -1; done
  echo "=== batch 8"; python tools/pair_bench.py B=8 2>/dev/null ) > $O/pair_bench.txt 2>&1
( for q in 0 1; do echo "=== VITAE_WS64Q=$q"; VITAE_WS64Q=$q python tools/ws64_phase_probe.py B=4 2>&1 | grep -v amdgpu.ids; done ) > $O/ws64_phase.txt 2>&1
for b in 4 8 32; do python tools/epi_tiles.py B=$b 2>&1 | grep -v amdgpu > $O/epi_tiles_b$b.txt; done
./build/probes/chain_rate > $O/chain_rate.txt 2>&1
python tools/optim_bench.py 2>&1 | grep -v amdgpu > $O/optim.txt
for w in "dec.fc1" "decoder" "decoder,enc.proj" ""; do VITAE_W2="$w" python tools/w2_parity.py 2>/dev/null | tail -1; done > $O/w2_parity.txt
python tools/ln_bench.py > $O/layernorm.txt 2>&1
grep real $O/bench_default.log; cut -c1-300 $O/bench_line.json
