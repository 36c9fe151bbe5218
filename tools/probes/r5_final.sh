# end of round 5: smoke(), the whole GPU suite, the profile refresh, the tile x form table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/probes/gpu_tests_full.sh
bash tools/probes/refresh_profiles_r5.sh > gpurun_out/final5/refresh.log 2>&1
timeout 900 python tools/bt_bench.py step forms=fwd,dgrad,wgrad tiles=5,4,6,3,0,-2,-1 > gpurun_out/final5/forms_raw.txt 2>&1; python tools/forms_table.py gpurun_out/final5/forms_raw.txt > gpurun_out/final5/bt_forms_table.txt
tail -3 gpurun_out/final5/bt_forms_table.txt; cut -c1-600 gpurun_out/final5/bench_line.json
