# kernel-by-kernel timeline of one replayed step (rocprofv3 --kernel-trace), fused and unfused
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2b/tr_fused -- python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 --profile-steps 0 > gpurun_out/r2b/tr_fused.log 2>&1
python tools/timeline.py $(ls gpurun_out/r2b/tr_fused/*/*kernel_trace.csv | head -1) 20 > gpurun_out/r2b/timeline_fused.txt 2>&1
VITAE_FUSE_MLP=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2b/tr_unfused -- python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 --profile-steps 0 > gpurun_out/r2b/tr_unfused.log 2>&1
python tools/timeline.py $(ls gpurun_out/r2b/tr_unfused/*/*kernel_trace.csv | head -1) 20 > gpurun_out/r2b/timeline_unfused.txt 2>&1
rm -rf gpurun_out/r2b/tr_fused gpurun_out/r2b/tr_unfused
tail -2 gpurun_out/r2b/tr_fused.log | cut -c1-200
