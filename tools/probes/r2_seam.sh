cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/seam
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seam/tr -- python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 --profile-steps 0 > gpurun_out/seam/tr.log 2>&1
f=$(ls gpurun_out/seam/tr/*/*kernel_trace.csv | head -1)
python tools/seam.py $f 30 31 40 > gpurun_out/seam/seam.txt 2>&1
python tools/timeline.py $f 30 > gpurun_out/seam/timeline.txt 2>&1
rm -rf gpurun_out/seam/tr
tail -1 gpurun_out/seam/tr.log | cut -c1-300
