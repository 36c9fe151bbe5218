# after the last planner refit: the form table at batch 32, a batch-32 / batch-4 bench pair, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/chk
timeout 900 python tools/bt_bench.py step forms=fwd,dgrad,wgrad tiles=5,4,3,0,-2,-1 > gpurun_out/chk/forms_raw.txt 2>&1; python tools/forms_table.py gpurun_out/chk/forms_raw.txt > gpurun_out/chk/forms.txt
tail -30 gpurun_out/chk/forms.txt
timeout 300 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline > gpurun_out/chk/bench_b4.txt 2>&1; tail -c 600 gpurun_out/chk/bench_b4.txt
timeout 300 python bench.py --batch 32 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/chk/bench_b32.txt 2>&1; tail -c 400 gpurun_out/chk/bench_b32.txt
bash tools/probes/gpu_tests_full.sh
