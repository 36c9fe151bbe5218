"""Seam between two replayed steps in a rocprofv3 kernel trace: the last kernels of step i and the first of step i+1, absolute
order across queues (is the first main-chain kernel held back by anything?)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'random_masking' in r['Kernel_Name']]
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:48]
for st in [int(a) for a in sys.argv[2:]] or [12, 13]:
    i = idx[st]
    t0 = int(rows[i]['Start_Timestamp'])
    print(f'--- step {st}: times relative to its random_masking start (us)')
    for r in rows[i - 14:i + 12]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%9.1f -> %9.1f  (%6.1f) q%s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r['Queue_Id'], short(r['Kernel_Name'])))
