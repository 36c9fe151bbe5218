# gap between the last kernel of a replayed step and the first kernel of the next one (kernel trace of the bench command)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6/st_gap -- python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/r6/gap.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r6/st_gap/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# a step starts at step_prologue_kernel
idx = [i for i, r in enumerate(rows) if 'step_prologue_kernel' in r[2]]
gaps, spans, inner = [], [], []
for a, b in zip(idx[:-1], idx[1:]):
    last_end = max(r[1] for r in rows[a:b])
    gaps.append((rows[b][0] - last_end) / 1e3)
    spans.append((rows[b][0] - rows[a][0]) / 1e3)
    inner.append(sum(max(0, rows[i + 1][0] - rows[i][1]) for i in range(a, b - 1)) / 1e3)
gaps, spans, inner = gaps[10:], spans[10:], inner[10:]
med = lambda v: sorted(v)[len(v) // 2]
print(f'steps {len(gaps)}: step period median {med(spans):.1f} us; gap last kernel -> next step\'s first kernel: median {med(gaps):.1f} us (min {min(gaps):.1f}, max {max(gaps):.1f}); gaps between nodes inside a step: {med(inner):.1f} us in total')
PY
rm -rf gpurun_out/r6/st_gap
