"""Per-step kernel time table from a rocprofv3 --kernel-trace --stats run of bench.py (steps = loss_finalize_kernel calls)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = [int(r['Calls']) for r in rows if 'loss_finalize_kernel' in r['Name']][0]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = 0
for r in rows:
    if 'spin' in r['Name']:
        continue
    t = int(r['TotalDurationNs']) / n / 1e3
    tot += t
    if t > thr:
        print(re.sub(r'\(anonymous namespace\)::', '', r['Name'])[:64].ljust(64), '%6.1f %8.1f %7.1f' % (int(r['Calls']) / n, t, float(r['AverageNs']) / 1e3))
print('total us/step', round(tot, 1), 'steps', n)
