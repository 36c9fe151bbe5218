cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "grouped or one_chain or recurring or two_plane or bench_workload" 2>&1 | grep -E "passed|failed|rror" | head -5
cat > /tmp/p8.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
import bench
args = bench.parse(['--no-extra', '--no-cpu-baseline'])
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
r = bench.other_config_point(args, dev, int(sys.argv[1]))
print(sys.argv[1], r['value'], r['ms_per_step'])
PY
run() { echo "$1 b$2: $(env $1 python bench.py --batch $2 --no-extra --no-cpu-baseline --steps 40 --warmup 10 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"; }
run X=0 8; run X=0 4
for i in 1 2; do echo "cfg4 default $(python /tmp/p8.py 4 2>&1 | tail -1)"; echo "cfg4 grouped(one queue) $(VITAE_SIDE_STREAMS=none VITAE_SIDE_MIN_ROWS=1 python /tmp/p8.py 4 2>&1 | tail -1)"; done
run VITAE_SIDE_MIN_ROWS=3500 16; run VITAE_SIDE_MIN_ROWS=2500 16
