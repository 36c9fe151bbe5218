import csv, sys, re, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'random_masking' in r['Kernel_Name']]
print('steps', len(idx))
st = int(sys.argv[2]) if len(sys.argv) > 2 else 30
seg = rows[idx[st]:idx[st+1]]
t0 = int(seg[0]['Start_Timestamp'])
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:60]
qs = collections.Counter(r['Queue_Id'] for r in seg)
print('queues', qs, 'span', (int(seg[-1]['End_Timestamp'])-t0)/1e3)
prev_end = {}
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r['Queue_Id']
    gap = (s - prev_end.get(q, s))/1e3
    prev_end[q] = e
    print('%8.1f %6.1f gap %5.1f q%s %s grid %s' % ((s-t0)/1e3, (e-s)/1e3, gap, q, short(r['Kernel_Name']), r['Grid_Size_X']))
