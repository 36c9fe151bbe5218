cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/tl; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 5 > $O/stats.log 2>&1
python tools/timeline.py $(ls $O/stats/*/*kernel_trace.csv | head -1) 25 > $O/timeline.txt 2>&1
rm -rf $O/stats
head -3 $O/timeline.txt
