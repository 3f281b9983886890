#!/usr/bin/env python3
"""per-kernel means of the counters collected by profiles/pmc_ab.sh"""
import csv, glob, os, sys, collections, re
for so in sys.argv[1:]:
    n = os.path.basename(so)[:-3]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/pab_%s_*/**/*counter_collection.csv' % n, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'lsx_sweep' not in k:
                continue
            m = re.search(r'<(-?\d+), ', k)
            acc[m.group(1) if m else k][row['Counter_Name']].append(float(row['Counter_Value']))
    print('==', n)
    for cls in sorted(acc):
        c = {k: sum(v) / len(v) for k, v in acc[cls].items()}
        w = c.get('SQ_WAVES', 1)
        line = 'class %3s waves %8d' % (cls, w)
        for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_WAVE_CYCLES'):
            if k in c:
                line += ' %s/w %.0f' % (k[3:].replace('INSTS_', ''), c[k] / w)
        print(line)
        line = '          '
        for k in ('SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY'):
            if k in c:
                line += ' %s %.3g' % (k[3:], c[k])
        print(line)
