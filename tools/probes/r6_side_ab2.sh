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
run() { echo "$1 b$2: $(env $1 python bench.py --batch $2 --no-extra --no-cpu-baseline --steps 40 --warmup 10 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"; }
runp() { echo "$1 $(env $1 python /tmp/p8.py $2 2>&1 | tail -1)"; }
{
for b in 8 16; do for v in all none wside; do run VITAE_SIDE_STREAMS=$v $b; done; done
for w in 4 5 p8; do for v in all none; do runp VITAE_SIDE_STREAMS=$v $w; done; done
run VITAE_OPT_IN_BACKWARD=0 4
run VITAE_OPT_IN_BACKWARD=1 4
run VITAE_OPT_IN_BACKWARD=0 8
run VITAE_OPT_IN_BACKWARD=1 8
} 2>&1 | tee gpurun_out/r6/side_ab2.txt
