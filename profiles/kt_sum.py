#!/usr/bin/env python3
"""per-kernel time of the big launches in a rocprofv3 kernel trace (profiles/kt_sum.py <trace.csv> <calls>)"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ncall = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
def short(n):
    m = re.search(r'(lsx_sweep_(?:rs_)?kernel(?:_all|_parabolic|_par)?<[^>]*>)', n)
    if m: return m.group(1).replace(' ', '')
    m = re.search(r'(k_\w+(<[^>]*>)?)', n); return m.group(1) if m else n[:30]
per = collections.defaultdict(list)
for r in rows:
    g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    per[short(r['Kernel_Name'])].append((g, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r['VGPR_Count']))
tot = 0
for k, v in sorted(per.items()):
    big = max(g for g, *_ in v)
    sel = [x for x in v if x[0] >= big / 40]
    ms = sum(x[1] for x in sel) / ncall
    if 'sweep' in k or 'fast' in k or 'finish' in k or 'stat' in k: tot += ms
    print('%-42s %4d launches  %.3f ms per call   vgpr %s' % (k, len(sel), ms, sel[0][2]))
print('sum of sweep + fast + finish + stat_equil: %.3f ms per call' % tot)
