"""Per-(kernel, grid size) averages of a rocprofv3 --pmc run: python tools/pmc_by_grid.py <dir> [name filter]"""
import csv, glob, os, sys, collections
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:48]
        if flt not in name:
            continue
        key = (name, r.get('Grid_Size') or r.get('Grid_Size_X'))
        agg[key][r['Counter_Name']] += float(r['Counter_Value'])
        seen[key].add(r.get('Dispatch_Id'))
    for key, c in sorted(agg.items()):
        n = len(seen[key])
        print(f'{key[0]:50s} grid={key[1]:>8s} n={n:4d} ' + ' '.join(f'{k}={v / n:.4g}' for k, v in sorted(c.items())))
