cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
cat > /tmp/cfg4.py <<'PY'
import argparse, sys, time, torch
sys.path.insert(0, '.')
import bench
args = bench.parse(['--no-extra', '--no-cpu-baseline'])
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
which = int(sys.argv[1])
print(bench.other_config_point(args, dev, which))
PY
for w in 4 5; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/st_cfg$w -- python /tmp/cfg4.py $w > gpurun_out/r6/cfg$w.log 2>&1
grep "value" gpurun_out/r6/cfg$w.log | cut -c1-300
python tools/prof_summary.py gpurun_out/r6/st_cfg$w 20 2>&1 | head -32 > gpurun_out/r6/kernels_cfg$w.txt
rm -rf gpurun_out/r6/st_cfg$w
cat gpurun_out/r6/kernels_cfg$w.txt
done
