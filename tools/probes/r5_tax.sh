# Round 5: where do the GEMMs of a compute-bound step lose 1.4-1.7x against their stand-alone durations?
# Kernel traces of the SAME launches (a) replayed from the step graph, (b) eager with the side branches, (c) eager with every
# branch on the main stream, (d) eager with a device synchronise after every launch (each kernel alone on an idle chip, operands
# as warm as its producer left them), and counter passes (clock = GRBM_GUI_ACTIVE / duration; L2 hit rate; MFMA busy) of (c).
# CFG="32:16:b32" (batch:patch:tag) selects the configuration.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CFG=${CFG:-32:16:b32}; B=${CFG%%:*}; r=${CFG#*:}; P=${r%%:*}; tag=${r#*:}
O=gpurun_out/r5tax_$tag; mkdir -p $O
COMMON="--batch $B --patch $P --steps 12 --warmup 3 --no-extra --no-cpu-baseline --profile-steps 0"
NOSIDE="VITAE_WGRAD_GROUP_SIDE=0 VITAE_PREDICTOR_SIDE=0 VITAE_OPT_IN_BACKWARD=0 VITAE_WGRAD_SIDE=0"
run() { name=$1; shift; env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/$name -- python bench.py $COMMON $EXTRA > $O/$name.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/$name.log | tail -1; }
EXTRA="" run graph X=1
EXTRA="--no-graph" run eager X=1
EXTRA="--no-graph" run eager_noside $NOSIDE
EXTRA="--no-graph" run alone $NOSIDE VITAE_SYNC_LAUNCHES=1
pmc() { name=$1; shift; env $NOSIDE rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python bench.py --batch $B --patch $P --steps 3 --warmup 1 --no-extra --no-cpu-baseline --profile-steps 0 --no-graph > $O/$name.log 2>&1; }
pmc pmc_clk GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS
pmc pmc_l2 TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES
python tools/in_step_tax.py $O > $O/in_step_tax.txt 2>&1
python tools/pmc_table.py $O/pmc_clk $O/pmc_l2 > $O/pmc_sq.txt 2>&1
for d in graph eager eager_noside alone pmc_clk pmc_l2; do f=$(ls $O/$d/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${d}_trace.csv; f=$(ls $O/$d/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && gzip -c $f > $O/${d}_counters.csv.gz; rm -rf $O/$d; done
gzip -f $O/*_trace.csv
head -60 $O/in_step_tax.txt
