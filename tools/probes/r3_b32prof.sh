cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/b32p; mkdir -p $O
B=${1:-32}; P=${2:-16}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python bench.py --batch $B --patch $P --steps 10 --warmup 4 --no-extra --no-cpu-baseline --profile-steps 0 > $O/st.log 2>&1
python tools/prof_summary.py $O/st 60 > $O/kernels_b${B}_p$P.txt 2>&1
rm -rf $O/st
head -45 $O/kernels_b${B}_p$P.txt
