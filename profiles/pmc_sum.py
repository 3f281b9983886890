#!/usr/bin/env python3
"""per-kernel sums of rocprofv3 counter-collection CSVs (profiles/pmc_sum.py dir ...): one line per (kernel, counter),
summed over dispatches and divided by the number of dispatches"""
import csv, collections, glob, os, re, sys
def short(n):
    m = re.search(r'(lsx_sweep_(?:rs_)?kernel(?:_all)?<[^>]*>)', n)
    if m: return m.group(1).replace(' ', '')
    m = re.search(r'(k_\w+(<[^>]*>)?)', n); return m.group(1) if m else n[:30]
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    nd = collections.defaultdict(set)
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r['Kernel_Name'])
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            nd[k].add(r['Dispatch_Id'])
    for k in sorted(acc):
        if not ('fast' in k or 'sweep' in k): continue
        n = len(nd[k])
        print('%-44s %3d dispatches: ' % (k, n) + '  '.join('%s %.4g' % (c.replace('_sum', ''), v / n) for c, v in sorted(acc[k].items())))
