"""Table of `tools/bt_bench.py step forms=fwd,dgrad,wgrad tiles=5,4,3,0,-2,-1` output: one row per (shape, form), one column per tile family
(us), the best marked, and the planner's pick against the best: python tools/forms_table.py <raw output>"""
import re, sys
rows, cur = {}, None
for l in open(sys.argv[1]):
    if l.startswith('--- '):
        cur = l[4:].strip()
        continue
    m = re.match(r'(\w+)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) tile\s+(\S+)->\s*(-?\d+) split=\s*(\d+)\s+([\d.]+) us', l)
    if m and cur:
        rows.setdefault((cur, m.group(1)), {})[m.group(5)] = (float(m.group(8)), int(m.group(6)), int(m.group(7)), '!!!' in l)
tiles = ['ws64', 'ws128', 'ws128x256', '128x128', '256x256', '64-row']
print(f'{"shape":<16}{"form":<6}' + ''.join(f'{t:>10}' for t in tiles) + f'{"auto":>10}  auto/best (auto tile, split)')
worst = 0.0
for (name, form), d in rows.items():
    best = min(v[0] for k, v in d.items() if k != 'auto' and not v[3])
    a = d.get('auto')
    cells = ''.join((f'{d[t][0]:>9.1f}' + ('*' if d[t][0] == best else '!' if d[t][3] else ' ')) if t in d else f'{"":>10}' for t in tiles)
    tail = f'{a[0]:>9.1f}   {a[0] / best:4.2f} ({a[1]}, s{a[2]})' if a else ''
    if a:
        worst = max(worst, a[0] / best)
    print(f'{name:<16}{form:<6}{cells}{tail}')
print(f'# worst auto/best {worst:.2f}')
