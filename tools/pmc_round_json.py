"""profiles/roundN_gemm_traffic.json from the counter passes of one round (tools/probes/refresh_profiles_r5.sh):
    python tools/pmc_round_json.py <FETCH pass> <WRITE pass> <SQ pass (pmc_clk)> <kernel_stats.csv of the default bench command> > profiles/round5_gemm_traffic.json
Per kernel (template arguments kept): memory-side bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction of
MI355X_MICROARCH.md section HBM), the rocprofv3 mean launch duration in graph replays, and the MFMA utilisation
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)  (busy cycles are 32 per v_mfma_f32_32x32x16_bf16, summed over SIMDs)."""
import csv, glob, gzip, json, os, re, sys, collections


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n).replace('vglds::', '')
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    return n.split('(')[0][:64]


def counters(path):
    fs = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True)
    agg, cnt, dur = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(lambda: collections.defaultdict(set)), collections.defaultdict(lambda: [0.0, 0])
    for f in fs:
        fh = gzip.open(f, 'rt') if f.endswith('.gz') else open(f)
        seen = set()
        for r in csv.DictReader(fh):
            n = short(r['Kernel_Name'])
            agg[n][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[n][r['Counter_Name']].add(r['Dispatch_Id'])
            if r['Dispatch_Id'] not in seen and r.get('End_Timestamp'):
                seen.add(r['Dispatch_Id'])
                dur[n][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
                dur[n][1] += 1
    return {n: ({c: v / len(cnt[n][c]) for c, v in cs.items()}, dur[n][0] / max(1, dur[n][1]), max(len(s) for s in cnt[n].values())) for n, cs in agg.items()}


fetch, write, sq = counters(sys.argv[1]), counters(sys.argv[2]), counters(sys.argv[3])
stats = {}
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
    for r in csv.DictReader(open(sys.argv[4])):
        stats[short(r['Name'])] = float(r['AverageNs']) / 1e3
TAGS = {'ws64_pair': 'gemm_ws64_pair_kernel', 'ws64': 'gemm_ws64_kernel', 'ws128': 'gemm_ws_kernel', 'bt256': 'gemm_bt_kernel<256', 'bt128': 'gemm_bt_kernel<128',
        'bt_group': 'gemm_bt_wgrad_group_kernel', 'attn': 'attn_'}
kern = {}
for n in sorted(set(fetch) | set(write) | set(sq)):
    e = {}
    if n in fetch and 'FETCH_SIZE' in fetch[n][0]:
        e['fetch_kb'] = round(fetch[n][0]['FETCH_SIZE'], 1)
    if n in write and 'WRITE_SIZE' in write[n][0]:
        e['write_kb'] = round(write[n][0]['WRITE_SIZE'], 1)
    if 'fetch_kb' in e and 'write_kb' in e:
        e['bytes_per_launch'] = int((2 * e['fetch_kb'] + e['write_kb']) * 1024)
    if n in sq:
        c, d, k = sq[n]
        e['launches_in_pass'] = k
        e['us_in_counter_pass'] = round(d, 2)
        if d and c.get('SQ_VALU_MFMA_BUSY_CYCLES'):
            e['mfma_util'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * d * 2400.0), 4)
            e['mfma_insts'] = int(c.get('SQ_INSTS_MFMA', 0))
        if c.get('SQ_WAVE_CYCLES'):
            e['wait_lds_frac'] = round(c.get('SQ_WAIT_INST_LDS', 0) / c['SQ_WAVE_CYCLES'], 4)
            e['wave_waiting_frac'] = round(c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES'], 4)
    if n in stats:
        e['rocprof_avg_launch_us_graph_replay'] = round(stats[n], 2)
    if e:
        kern[n] = e


def fam(prefix, key, weight='launches_in_pass'):
    sel = [(v.get(weight, 1), v[key]) for n, v in kern.items() if n.startswith(prefix) and key in v]
    w = sum(a for a, _ in sel)
    return sum(a * b for a, b in sel) / w if w else None


out = {'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | (SQ + GRBM) of `python bench.py --batch 4 --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-extra --no-graph` '
                 '(separate passes, no trace domains beside them; the SQ pass with the side branches on the main stream) and the kernel stats of the default bench command; tools/probes/refresh_profiles_r5.sh',
       'correction': 'bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-B requests as 64 B, MI355X_MICROARCH.md section HBM; WRITE_SIZE at face value); '
                     'mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * duration * 2.4 GHz)',
       'kernels': kern, 'bytes_per_launch': {}, 'rocprof_avg_launch_us': {}, 'mfma_busy_frac': {}}
for tag, pre in TAGS.items():
    b, u, m = fam(pre, 'bytes_per_launch'), fam(pre, 'rocprof_avg_launch_us_graph_replay'), fam(pre, 'mfma_util')
    if b is not None:
        out['bytes_per_launch'][tag] = int(b)
    if u is not None:
        out['rocprof_avg_launch_us'][tag] = round(u, 2)
    if m is not None:
        out['mfma_busy_frac'][tag] = round(m, 4)
print(json.dumps(out, indent=1))
