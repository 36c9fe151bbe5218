cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/tl; mkdir -p $O
B=${1:-4}; P=${2:-16}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --batch $B --patch $P --no-cpu-baseline --no-extra --steps 12 --warmup 4 --profile-steps 0 > $O/stats.log 2>&1
python tools/timeline.py $(ls $O/stats/*/*kernel_trace.csv | head -1) ${3:-20} > $O/timeline_b${B}_p$P.txt 2>&1
rm -rf $O/stats
head -3 $O/timeline_b${B}_p$P.txt
