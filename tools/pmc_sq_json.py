"""MFMA-busy share of the GEMM + attention kernels of one step from a rocprofv3 SQ pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA ...):
    python tools/pmc_sq_json.py <pass dir or counters.csv[.gz]> > profiles/round5_pmc_sq_b8.json"""
import csv, glob, gzip, json, os, re, sys, collections
path = sys.argv[1]
fs = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True)
busy = collections.defaultdict(float); dur = collections.defaultdict(float); n = collections.Counter(); seen = set()
for f in fs:
    fh = gzip.open(f, 'rt') if f.endswith('.gz') else open(f)
    for r in csv.DictReader(fh):
        k = re.sub(r'\(anonymous namespace\)::|^void |vglds::|_ZN12_GLOBAL__N_1\d+', '', r['Kernel_Name']).split('(')[0][:64]
        if not ('gemm' in k or 'attn_' in k):
            continue
        if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
            busy[k] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            n[k] += 1
tb, td = sum(busy.values()), sum(dur.values())
print(json.dumps({'source': 'rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS of bench.py at this batch (eager, side branches on the main stream)',
                  'definition': 'busy cycles (32 per v_mfma_f32_32x32x16_bf16, summed over SIMDs) / (1024 SIMDs x kernel duration x 2.4 GHz)',
                  'gemm_attn_mfma_busy_frac': round(tb / (1024 * td * 2400.0), 4) if td else None,
                  'per_kernel': {k: {'launches': n[k], 'us': round(dur[k] / n[k], 2), 'mfma_busy_frac': round(busy[k] / (1024 * dur[k] * 2400.0), 4)} for k in sorted(dur, key=lambda k: -dur[k])}}, indent=1))
