cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
run() { echo "$1: $(env $1 python bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"; }
for rep in 1 2; do
run X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1000
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run ROC_USE_FGS_KERNARG=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run GPU_MAX_HW_QUEUES=8
run ROC_AQL_QUEUE_SIZE=65536
done 2>&1 | tee gpurun_out/r6/rt_knobs.txt
