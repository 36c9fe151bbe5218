cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2j/st -- python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 --profile-steps 0 --batch 32 > gpurun_out/r2j/b32.log 2>&1
cp $(ls gpurun_out/r2j/st/*/*kernel_stats.csv | head -1) gpurun_out/r2j/b32_kernel_stats.csv
rm -rf gpurun_out/r2j/st
