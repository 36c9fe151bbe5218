cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/tl; mkdir -p $O
for cfg in "32:16:b32" "4:8:p8"; do
  B=${cfg%%:*}; r=${cfg#*:}; P=${r%%:*}; tag=${r#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$tag -- python bench.py --batch $B --patch $P --steps 30 --warmup 5 --no-extra --no-cpu-baseline --profile-steps 0 > $O/st_$tag.log 2>&1
  python tools/timeline.py $(ls $O/st_$tag/*/*kernel_trace.csv | head -1) 30 > $O/timeline_$tag.txt 2>&1
  python tools/prof_summary.py $O/st_$tag 25 > $O/kernels_$tag.txt 2>&1
  rm -rf $O/st_$tag
done
