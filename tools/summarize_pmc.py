"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel family: python tools/summarize_pmc.py <dir> [...]"""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
            agg[name][r['Counter_Name']] += float(r['Counter_Value'])
            key = (r.get('Dispatch_Id'), name)
            if key not in seen:
                seen.add(key); cnt[name] += 1
        print('==', f)
        rows = sorted(agg.items(), key=lambda kv: -max(kv[1].values()))[:14]
        for name, c in rows:
            print(f'{name:62s} n={cnt[name]:5d} ' + ' '.join(f'{k}={v/cnt[name]:.4g}' for k, v in sorted(c.items())))
