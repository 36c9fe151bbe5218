cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/tlx3; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python bench.py --precision fp32x3 --steps 30 --warmup 5 --no-extra --no-cpu-baseline --profile-steps 0 > $O/st.log 2>&1
python tools/timeline.py $(ls $O/st/*/*kernel_trace.csv | head -1) 30 > $O/timeline.txt 2>&1
python tools/prof_summary.py $O/st 8 > $O/kernels.txt 2>&1
rm -rf $O/st
