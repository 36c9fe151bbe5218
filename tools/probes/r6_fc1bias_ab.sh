cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
cat > /tmp/p8.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
import bench
args = bench.parse(['--no-extra', '--no-cpu-baseline'])
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
r = bench.patch8_point(args, dev)
print('p8', r['value'], r['ms_per_step'])
PY
for i in 1 2; do for v in 0 1; do
echo "GROUP=$v b32: $(VITAE_FC1_BIAS_GROUP=$v python bench.py --batch 32 --no-extra --no-cpu-baseline --steps 30 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
echo "GROUP=$v $(VITAE_FC1_BIAS_GROUP=$v python /tmp/p8.py 2>&1 | tail -1)"
done; done
python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "b32 or batch32 or grouped or group or patch8 or p8" 2>&1 | tail -3
