"""MFMA-utilisation and LDS counters per kernel from rocprofv3 --pmc passes of bench.py (tools/probes/r5_tax.sh, r5_pmc.sh):
    python tools/pmc_table.py <pass dir or *_counters.csv.gz> ...
Per kernel (template arguments kept) the mean per launch of every collected counter and the derived shares the north star asks for:
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES   (matrix pipe busy while the shader engines are busy)
  wait_lds  = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES,  wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES,  issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_conf  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE,   l2_hit = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  ghz       = GRBM_GUI_ACTIVE / duration (the shader clock during the dispatch)."""
import csv, glob, gzip, json, os, re, sys, collections

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(lambda: [0.0, 0])
for a in sys.argv[1:]:
    if a.startswith('--'):
        continue
    fs = [a] if os.path.isfile(a) else glob.glob(os.path.join(a, '**', '*counter_collection.csv'), recursive=True)
    for f in fs:
        fh = gzip.open(f, 'rt') if f.endswith('.gz') else open(f)
        seen = set()
        for r in csv.DictReader(fh):
            n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            n = re.sub(r'^void ', '', n).replace('vglds::', '').split('(')[0][:64]
            c = r['Counter_Name']
            agg[n][c] += float(r['Counter_Value'])
            cnt[n][c].add((f, r['Dispatch_Id']))
            if (f, r['Dispatch_Id']) not in seen and r.get('End_Timestamp'):
                seen.add((f, r['Dispatch_Id']))
                dur[n][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
                dur[n][1] += 1
rows = {}
for n, cs in agg.items():
    m = {c: v / len(cnt[n][c]) for c, v in cs.items()}
    d = dur[n][0] / dur[n][1] if dur[n][1] else 0.0
    g = lambda k: m.get(k, 0.0)
    der = {'launches': max(len(s) for s in cnt[n].values()), 'us': d}
    if g('SQ_BUSY_CYCLES'):
        der['mfma_busy'] = g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES')
    if g('SQ_WAVE_CYCLES'):
        der['wait_lds'] = g('SQ_WAIT_INST_LDS') / g('SQ_WAVE_CYCLES')
        der['wait_any'] = g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES')
        der['issue'] = g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES')
    if g('SQ_LDS_IDX_ACTIVE'):
        der['lds_conf'] = g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')
    if g('TCC_HIT_sum') + g('TCC_MISS_sum'):
        der['l2_hit'] = g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum'))
    if d and g('GRBM_GUI_ACTIVE'):
        der['ghz'] = g('GRBM_GUI_ACTIVE') / (d * 1e3)
    rows[n] = (der, m)
order = sorted(rows, key=lambda n: -rows[n][0]['us'] * rows[n][0]['launches'])
if '--json' in sys.argv:
    print(json.dumps({n: {**{k: round(v, 5) for k, v in rows[n][0].items()}, **{k: v for k, v in rows[n][1].items()}} for n in order[:40]}))
    sys.exit(0)
for n in order[:40]:
    der, m = rows[n]
    print(f'{n:64s} ' + ' '.join(f'{k}={v:.4g}' for k, v in der.items()))
    print(' ' * 8 + ' '.join(f'{k}={v:.4g}' for k, v in sorted(m.items())))
