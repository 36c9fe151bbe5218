"""Table of tools/bt_bench.py 'step' output: one row per shape, one column per tile (us), best marked; optional library column
from profiles/round2_gemm_vs_library.txt."""
import re, sys
lib = {}
try:
    for l in open('profiles/round2_gemm_vs_library.txt'):
        m = re.match(r'(B\d+ .*?)\s+(\d+)\s+(\d+)\s+(\d+) \|\s+([\d.]+)', l)
        if m:
            lib[m.group(1).strip()] = float(m.group(5))
except OSError:
    pass
rows, cur = {}, None
for l in open(sys.argv[1]):
    if l.startswith('--- '):
        cur = l[4:].strip(); rows.setdefault(cur, {})
        continue
    m = re.match(r'(\w+)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) tile\s+(\S+)->\s*(-?\d+) split=\s*(\d+)\s+([\d.]+) us', l)
    if m and cur:
        rows[cur].setdefault(m.group(1), {})[m.group(5)] = float(m.group(8))
tiles = ['256x256', '128x128', '64-row', 'auto']
print(f'{"shape":<18}{"form":<6}' + ''.join(f'{t:>9}' for t in tiles) + f'{"library":>9}{"best/lib":>9}')
for name, forms in rows.items():
    for form, d in forms.items():
        best = min(v for k, v in d.items() if k != 'auto')
        lb = lib.get(name) if form == 'fwd' else None
        print(f'{name:<18}{form:<6}' + ''.join((f'{d[t]:>8.1f}' + ('*' if d[t] == best else ' ')) if t in d else f'{"":>9}' for t in tiles)
              + (f'{lb:>9.1f}{best / lb:>9.2f}' if lb else ''))
