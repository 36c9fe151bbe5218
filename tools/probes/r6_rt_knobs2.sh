cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
cat > /tmp/p8.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
import bench
args = bench.parse(['--no-extra', '--no-cpu-baseline'])
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
if sys.argv[1] == 'p8':
    r = bench.patch8_point(args, dev)
else:
    r = bench.other_config_point(args, dev, int(sys.argv[1]))
print(sys.argv[1], r['value'], r['ms_per_step'])
PY
run() { echo "$1 b$2: $(env $1 python bench.py --batch $2 --no-extra --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"; }
runp() { echo "$1 $(env $1 python /tmp/p8.py $2 2>&1 | tail -1)"; }
{
for q in 1 2 3; do run DEBUG_HIP_FORCE_GRAPH_QUEUES=$q 4; done
for q in 1 2; do run GPU_MAX_HW_QUEUES=$q 4; done
for b in 8 16 32; do run X=0 $b; run DEBUG_HIP_FORCE_GRAPH_QUEUES=1 $b; run DEBUG_HIP_FORCE_GRAPH_QUEUES=2 $b; done
for w in p8 4 5; do runp X=0 $w; runp DEBUG_HIP_FORCE_GRAPH_QUEUES=1 $w; runp DEBUG_HIP_FORCE_GRAPH_QUEUES=2 $w; done
} 2>&1 | tee gpurun_out/r6/rt_knobs2.txt
