cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for w in "dec.fc1" "" ; do
VITAE_W2="$w" python bench.py --batch 4 --no-cpu-baseline --no-extra --steps 60 --warmup 10 --profile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('W2=[$w] B=4', d['ms_per_step'], 'ms')"
done; done
VITAE_W2="" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/st_now2 -- python bench.py --batch 4 --steps 30 --warmup 5 --no-extra --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/r6/st_now2 8 | head -14
rm -rf gpurun_out/r6/st_now2
