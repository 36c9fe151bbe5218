"""Round 5: the same launches of one step under four conditions (tools/probes/r5_tax.sh).
    python tools/in_step_tax.py <dir with graph/ eager/ eager_noside/ alone/ [pmc_clk/ pmc_l2/]>
Per (kernel, grid): launches per step and the mean duration when the step is replayed from its graph, issued eagerly with the side
branches, eagerly with every branch on the main stream, and with a device synchronise after every launch ("alone": idle chip,
operands as warm as the producer left them); then the counter passes: duration while the counters are collected (dispatches
serialised by the profiler), shader clock = GRBM_GUI_ACTIVE / duration, L2 hit rate, MFMA-busy share."""
import csv, glob, gzip, os, re, sys, collections

root = sys.argv[1]
MIN_US = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0     # per-step total below which a kernel is not listed


def open_any(pat):
    fs = glob.glob(pat, recursive=True)
    if not fs:
        return None
    f = fs[0]
    return gzip.open(f, 'rt') if f.endswith('.gz') else open(f)


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = n.replace('vglds::', '')
    return n.split('(')[0][:58]


def trace(name):
    fh = open_any(f'{root}/{name}/**/*kernel_trace.csv') or open_any(f'{root}/{name}_trace.csv*')
    if fh is None:
        return None, 0
    rows = list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'random_masking' in r['Kernel_Name']]
    if len(idx) < 6:
        return None, 0
    lo, hi = idx[len(idx) // 2], idx[-1]            # second half of the run: past warm-up and graph capture
    nsteps = len(idx) - 1 - len(idx) // 2
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[lo:hi]:
        k = (short(r['Kernel_Name']), r['Grid_Size_X'], r.get('Grid_Size_Z', '1'))
        a = agg[k]
        a[0] += 1
        a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    span = (int(rows[hi]['Start_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e3 / nsteps
    return {k: (v[0] / nsteps, v[1] / v[0]) for k, v in agg.items()}, span


def counters(name):
    fh = open_any(f'{root}/{name}/**/*counter_collection.csv') or open_any(f'{root}/{name}_counters.csv*')
    if fh is None:
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(fh):
        gx = r.get('Grid_Size_X') or r.get('Grid_Size')
        k = (short(r['Kernel_Name']), gx)
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        d = r.get('Dispatch_Id')
        if d not in seen[k]:
            seen[k].add(d)
            if r.get('End_Timestamp'):
                dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    return {k: ({c: v / len(seen[k]) for c, v in cs.items()}, dur[k] / len(seen[k])) for k, cs in agg.items()}


names = ['graph', 'eager', 'eager_noside', 'alone']
T = {}
for n in names:
    T[n], span = trace(n)
    print(f'# {n}: {"missing" if T[n] is None else f"{span:.1f} us per step (trace span)"}')
clk, l2 = counters('pmc_clk'), counters('pmc_l2')
base = T['graph'] or T['eager']
keys = sorted(base, key=lambda k: -base[k][0] * base[k][1])
print(f'{"kernel":58s} {"grid":>7s} z  n/step | ' + ' '.join(f'{n[:9]:>9s}' for n in names) + ' | graph/alone |  pmc us  GHz  L2hit  mfma/busy  waitLDS/wave')
tot = {n: 0.0 for n in names}
for k in keys:
    n0, d0 = base[k]
    for n in names:
        if T[n] and k in T[n]:
            tot[n] += T[n][k][0] * T[n][k][1]
    if n0 * d0 < MIN_US:
        continue
    cols = []
    for n in names:
        cols.append(f'{T[n][k][1]:9.1f}' if T[n] and k in T[n] else f'{"-":>9s}')
    al = T['alone'][k][1] if T['alone'] and k in T['alone'] else None
    ratio = f'{d0 / al:11.2f}' if al else f'{"-":>11s}'
    # the counter files key the grid by its total thread count: match by name and x-extent (Grid_Size = x * y * z threads)
    pm = ''
    cands = [kk for kk in clk if kk[0] == k[0]]
    kk = next((c for c in cands if c[1] in (k[1], str(int(k[1]) * int(k[2] or 1)))), None)
    if kk:
        c, d = clk[kk]
        ghz = c.get('GRBM_GUI_ACTIVE', 0) / (d * 1e3) if d else 0
        busy = c.get('SQ_BUSY_CYCLES', 0)
        mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy if busy else 0
        wl = c.get('SQ_WAIT_INST_LDS', 0) / c['SQ_WAVE_CYCLES'] if c.get('SQ_WAVE_CYCLES') else 0
        hit = ''
        if kk in l2:
            h, m = l2[kk][0].get('TCC_HIT_sum', 0), l2[kk][0].get('TCC_MISS_sum', 0)
            hit = f'{h / (h + m):5.2f}' if h + m else ''
        pm = f'{d:7.1f} {ghz:5.2f} {hit:>5s} {mf:9.3f} {wl:9.3f}'
    print(f'{k[0]:58s} {k[1]:>7s} {k[2]:>2s} {n0:6.1f} | ' + ' '.join(cols) + f' | {ratio} | {pm}')
print('# kernel time per step (sum over all kernels, us): ' + ', '.join(f'{n} {tot[n]:.0f}' for n in names if T[n]))
