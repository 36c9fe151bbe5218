cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/st_tl -- python bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r6/tl.log 2>&1
python tools/timeline.py $(ls gpurun_out/r6/st_tl/*/*kernel_trace.csv | head -1) 30 > gpurun_out/r6/timeline_b4_q1.txt 2>&1
rm -rf gpurun_out/r6/st_tl
head -3 gpurun_out/r6/timeline_b4_q1.txt
