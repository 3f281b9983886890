#!/usr/bin/env python3
"""One formal-solution call of a rocprofv3 --kernel-trace run as a timeline: python3 profiles/timeline.py <dir with *kernel_trace.csv> [call]
(kernels between two consecutive k_gamma_finish launches; times relative to the end of the previous call's epilogue)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_trace.csv')[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_gamma_finish' in r['Kernel_Name']]
a, b = idx[which - 1], idx[which]
t0 = int(rows[a]['End_Timestamp'])
for r in rows[a + 1:b + 1]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:48]
    print('%-50s start %8.1f us  duration %8.1f us  stream %s' % (n, (int(r['Start_Timestamp']) - t0) / 1e3,
                                                                 (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Stream_Id')))
