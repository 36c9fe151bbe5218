cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/x3prof; mkdir -p $O
for prec in fp32x3 fp32; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$prec -- python bench.py --precision $prec --steps 20 --warmup 5 --no-extra --no-cpu-baseline --profile-steps 0 > $O/st_$prec.log 2>&1
python tools/prof_summary.py $O/st_$prec 25 > $O/kernels_$prec.txt 2>&1
python tools/timeline.py $(ls $O/st_$prec/*/*kernel_trace.csv | head -1) 20 > $O/timeline_$prec.txt 2>&1
rm -rf $O/st_$prec
grep -o '"ms_per_step": [0-9.]*' $O/st_$prec.log | tail -1
done
