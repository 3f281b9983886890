#!/usr/bin/env python3
"""Summarise rocprofv3 CSVs of profiles/collect.sh: per-kernel averages of every counter.
usage: python profiles/summarize.py gpurun_out/<tag> [kernel-substring] [min_grid]"""
import csv, glob, os, sys, collections, json

def main():
    tag = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else 'lsx_sweep_kernel'
    min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    out = {}
    for d in sorted(glob.glob(tag + '_pmc*')):
        for f in glob.glob(os.path.join(d, '*counter_collection.csv')):
            acc = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                if want not in row['Kernel_Name']:
                    continue
                if int(row.get('Grid_Size', 0)) < min_grid:
                    continue
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
            for k, v in acc.items():
                out[k] = dict(mean=sum(v) / len(v), n=len(v))
    kt = glob.glob(tag + '_kt/*kernel_trace.csv')
    if kt:
        durs = []
        for row in csv.DictReader(open(kt[0])):
            if want in row['Kernel_Name'] and int(row.get('Grid_Size', 0)) >= min_grid:
                durs.append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
        if durs:
            out['duration_ns'] = dict(mean=sum(durs) / len(durs), n=len(durs), min=min(durs), max=max(durs))
    print(json.dumps(out, indent=1))

if __name__ == '__main__':
    main()
