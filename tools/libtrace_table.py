"""Calibration table from a rocprofv3 --kernel-trace of tools/gemm_vs_library.py: per shape, the median kernel duration of the
vendor library's launches (torch.matmul -> hipBLASLt) and of ours (whatever family csrc/gemm_glds.hip's planner picked)."""
import csv, re, sys
sys.path.insert(0, '.')
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
runs, cur = [], None
for r in rows:
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    fam = 'ours' if ('gemm_glds' in n or 'gemm_bt' in n or 'gemm_ws' in n) else 'lib' if ('Cijk' in n or 'gemm' in n.lower() or 'Tensile' in n) else None
    if fam is None:
        continue
    if cur and cur[0] == n:
        cur[2].append(d)
    else:
        cur = [n, fam, [d]]
        runs.append(cur)
runs = [r for r in runs if len(r[2]) >= 50]
SHAPES = []
for B in (4, 8, 32):
    Me, Md = B * 2 * 55, B * 217
    SHAPES += [(f'B{B} enc qkv', Me, 2304, 768), (f'B{B} enc proj', Me, 768, 768), (f'B{B} enc fc1', Me, 3072, 768),
               (f'B{B} enc fc2', Me, 768, 3072), (f'B{B} dec qkv', Md, 1536, 512), (f'B{B} dec fc1', Md, 2048, 512),
               (f'B{B} dec fc2', Md, 512, 2048), (f'B{B} dec pred', Md, 16384, 512), (f'B{B} patch embed', B * 2 * 54, 768, 16384)]


def short(n):
    m = re.search(r'MT(\d+x\d+x\d+)', n)
    if m:
        return m.group(1)
    if 'gemm_ws64' in n:
        return 'ws 64x64'
    if 'gemm_ws_kernel' in n:
        return 'ws 128x128'
    m = re.search(r'gemm_bt_kernel<(\d+), (\d+)', n)
    if m:
        return f'bt {m.group(1)}x{m.group(2)}'
    m = re.search(r'gemm_glds_kernel<(\d+), (\d+)', n)
    if m:
        return f'{m.group(1)}x{m.group(2)}'
    return 'pipe 64x64' if 'pipe' in n else n[:20]


print('# GPU kernel durations (median of >= 200 back-to-back launches, 4 operand sets cycled; rocprofv3 --kernel-trace of tools/gemm_vs_library.py):')
print('# y[M,N] = x[M,K] W[N,K]^T, bf16 in; library = torch.matmul -> hipBLASLt (bf16 out), ours = vitae_gemm_glds with the planner\'s choice, fp32 result and bf16-only result.')
print(f'{"shape":<18}{"M":>6}{"N":>6}{"K":>6} | {"library us":>10} {"TF/s":>6} {"tile":>14} | {"ours us":>8} {"TF/s":>6} {"kernel":>12} | ours/library | ours with a bf16-only result: us, TF/s, ratio')
# the tool launches lib then ours per shape: pair them up in order
libs = [r for r in runs if r[1] == 'lib']
ours = [r for r in runs if r[1] == 'ours']
for i, (name, M, N, K) in enumerate(SHAPES):
    if i >= len(libs) or i >= len(ours):
        break
    med = lambda x: sorted(x)[len(x) // 2]
    d = ours[i][2]
    # (ours runs twice per shape with the same kernel: 210 launches with an fp32 result, then 210 with a bf16-only result — like for like
    # with the library, and what the step's qkv / fc1 launches write)
    o, o16 = (med(d[:len(d) // 2]), med(d[len(d) // 2:])) if len(d) >= 380 else (med(d), float('nan'))
    l = med(libs[i][2])
    fl = 2.0 * M * N * K
    print(f'{name:<18}{M:>6}{N:>6}{K:>6} | {l:>10.1f} {fl / l / 1e6:>6.0f} {short(libs[i][0]):>14} | {o:>8.1f} {fl / o / 1e6:>6.0f} {short(ours[i][0]):>12} | {o / l:.2f} | bf16 out {o16:>7.1f} {fl / o16 / 1e6:>6.0f} {o16 / l:.2f}')
