#!/usr/bin/env python3
"""per-kernel HBM bytes of one formal solution from two rocprofv3 counter passes (profiles/traffic_ab.sh):
    traffic_sum.py NAME NCOL dir_FETCH dir_WRITE
calls = launches of the Gamma epilogue; bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024"""
import csv, collections, glob, os, re, sys
def short(n):
    m = re.search(r'(lsx_sweep_(?:rs_)?kernel(?:_all)?<[^>]*>)', n)
    if m: return m.group(1).replace(' ', '')
    m = re.search(r'(k_\w+(<[^>]*>)?)', n); return m.group(1) if m else n[:30]
name, ncol = sys.argv[1], int(sys.argv[2])
tot = collections.defaultdict(lambda: [0.0, 0.0]); calls = [0, 0]
for i, d in enumerate(sys.argv[3:5]):
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] not in ('FETCH_SIZE', 'WRITE_SIZE'): continue
            if not any(x in r['Kernel_Name'] for x in ('lsx_sweep_', 'k_fast_', 'k_gamma_finish', 'k_build_optab')): continue
            tot[short(r['Kernel_Name'])][i] += float(r['Counter_Value'])
            if 'k_gamma_finish' in r['Kernel_Name']: calls[i] += 1
s = 0.0; sw = 0.0
print('== %s  (%d / %d calls counted, %d columns)' % (name, calls[0], calls[1], ncol))
for k in sorted(tot):
    f, w = tot[k][0] * 2 * 1024 / max(calls[0], 1), tot[k][1] * 1024 / max(calls[1], 1)
    s += f + w
    if 'lsx_sweep_' in k: sw += f + w
    print('  %-52s read %8.3f GB  write %8.3f GB  = %7.3f MB/column' % (k, f / 1e9, w / 1e9, (f + w) / ncol / 1e6))
print('  %s: sweeps %.3f MB/column, whole call %.3f MB/column' % (name, sw / ncol / 1e6, s / ncol / 1e6))
