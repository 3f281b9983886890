#!/usr/bin/env python3
"""loops of every kernel in a `hipcc -S --cuda-device-only` listing: instruction counts per loop body (vector, fp64, LDS, scratch, moves):
python3 profiles/isa_loops.py file.s [min_len]"""
import re, sys
s = open(sys.argv[1]).read()
minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ks = re.split(r'\n(_Z\w+):[^\n]*\n', s)
for i in range(1, len(ks), 2):
    name, body = ks[i], ks[i + 1].split('.section')[0]
    lines = body.split('\n')
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r'(\.LBB\d+_\d+):', l)] if m}
    loops = []
    for n, l in enumerate(lines):
        m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            loops.append((labels[m.group(1)], n))
    rows = []
    for a, b in loops:
        if b - a > minlen:
            seg = lines[a:b]
            cnt = lambda pat: sum(bool(re.match(pat, l)) for l in seg)
            rows.append('   loop @%d len %d: valu %d f64 %d trans %d ds %d vmem ld/st %d/%d scratch ld/st %d/%d mov %d waitcnt %d' % (
                a, b - a, cnt(r'\s+v_'), cnt(r'\s+v_\w+_f64'), cnt(r'\s+v_(rcp|exp|log|sqrt|rsq)'), cnt(r'\s+ds_'),
                cnt(r'\s+(global|buffer|flat)_load'), cnt(r'\s+(global|buffer|flat)_store'), cnt(r'\s+scratch_load'), cnt(r'\s+scratch_store'),
                cnt(r'\s+v_mov|\s+v_accvgpr'), cnt(r'\s+s_waitcnt')))
    if rows:
        print(name)
        print('\n'.join(rows))
