"""Idle gaps (no kernel of any queue running) inside the replayed steps of a rocprofv3 --kernel-trace of bench.py."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'loss_finalize_kernel' in r['Kernel_Name']]
tot = []
for st in range(12, min(len(idx) - 1, 40)):
    seg = rows[idx[st]:idx[st + 1]]
    t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
    cur = int(seg[0]['End_Timestamp']); idle = 0; big = 0
    for r in seg[1:]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > cur:
            idle += s - cur
            big += (s - cur) if s - cur > 25000 else 0
        cur = max(cur, e)
    tot.append(((t1 - t0) / 1e3, idle / 1e3, big / 1e3, len({r['Queue_Id'] for r in seg})))
print('steps', len(tot), 'span %.0f us  idle %.0f us  of which gaps > 25 us: %.0f us  queues %s' % (
    sum(t[0] for t in tot) / len(tot), sum(t[1] for t in tot) / len(tot), sum(t[2] for t in tot) / len(tot), sorted({t[3] for t in tot})))
