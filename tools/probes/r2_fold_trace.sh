cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fold
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fold/tr -- python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 --profile-steps 0 > gpurun_out/fold/tr.log 2>&1
f=$(ls gpurun_out/fold/tr/*/*kernel_trace.csv | head -1)
python tools/timeline.py $f 30 > gpurun_out/fold/timeline.txt 2>&1
rm -rf gpurun_out/fold/tr
sed -n 1,80p gpurun_out/fold/timeline.txt | cut -c1-130
