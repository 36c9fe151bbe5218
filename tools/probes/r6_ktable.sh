cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/st_b4 -- python bench.py --no-extra --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r6/b4q.log 2>&1
tail -1 gpurun_out/r6/b4q.log | cut -c1-200
python tools/prof_summary.py gpurun_out/r6/st_b4 6 2>&1 | head -70 > gpurun_out/r6/kernels_b4_q1.txt
rm -rf gpurun_out/r6/st_b4
cat gpurun_out/r6/kernels_b4_q1.txt
