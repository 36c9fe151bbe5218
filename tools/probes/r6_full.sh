cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6/gpu_tests_full.txt 2>&1
echo tests rc=$?
grep -n "Error\|FAILED\|passed\|failed" gpurun_out/r6/gpu_tests_full.txt | head -10
( time python bench.py ) > gpurun_out/r6/bench_default.log 2>&1
grep '^{' gpurun_out/r6/bench_default.log | tail -1 > gpurun_out/r6/bench_line.json
grep real gpurun_out/r6/bench_default.log
python - <<PY
import json
d=json.load(open('gpurun_out/r6/bench_line.json'))
print(d['value'], d['ms_per_step'])
c=d['config']
for k in ('also_batch8','also_batch32','also','also_fp32','also_fp32x3','also_epoch_loop','also_p8','also_cfg4','also_cfg5'):
    v=c.get(k,{}); print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','ms_per_iteration','step_frac_of_peak_reference_formulation','error') if kk in v})
print('roofline', {k:d['roofline'].get(k) for k in ('kernel','achieved','frac','avg_launch_us','traffic','frac_at_rocprof_duration')}, d['roofline'].get('encoder_attn_mlp_b8'))
print('cpu', d['cpu_baseline'])
p=d['parity']; print({k:p.get(k) for k in ('recon_rel_err','worst_total_loss_rel_err','worst_raw_edge_rel_err','worst_contr_rel_err','option_w2_decoder')})
PY
