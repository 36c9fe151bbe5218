cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2i/tr -- python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 --profile-steps 0 > gpurun_out/r2i/tr.log 2>&1
python tools/timeline.py $(ls gpurun_out/r2i/tr/*/*kernel_trace.csv | head -1) 20 > gpurun_out/r2i/timeline_slabk.txt 2>&1
rm -rf gpurun_out/r2i/tr
