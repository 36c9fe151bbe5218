cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
cat > /tmp/cfg4.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
import bench
args = bench.parse(['--no-extra', '--no-cpu-baseline'])
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
r = bench.other_config_point(args, dev, int(sys.argv[1]))
print(sys.argv[1], r['value'], r.get('ms_per_step'))
PY
for i in 1 2; do for v in 0 1; do echo "REST=$v: $(VITAE_PAIR_WS64_REST=$v python /tmp/cfg4.py 4 2>&1 | tail -1)"; done; done
python -m pytest tests -m gpu -x -q -k "pair or planner or bt_" 2>&1 | tail -3
