#!/usr/bin/env python3
"""vector registers live into the largest loop of a kernel in a `hipcc -S` listing (read in the loop body before any write to them, in
program order): the loop-carried and loop-invariant values -- python3 profiles/isa_livein.py file.s kernel-substring"""
import re, sys
s = open(sys.argv[1]).read()
ks = re.split(r'\n(_Z\w+):[^\n]*\n', s)
for i in range(1, len(ks), 2):
    if sys.argv[2] not in ks[i]:
        continue
    lines = ks[i + 1].split('.section')[0].split('\n')
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r'(\.LBB\d+_\d+):', l)] if m}
    loops = []
    for n, l in enumerate(lines):
        m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            loops.append((n - labels[m.group(1)], labels[m.group(1)], n))
    ln, a, b = max(loops)
    def regs(tok):
        out = []
        for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
            if m.group(3) is not None: out.append(int(m.group(3)))
            else: out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
        return out
    written, livein, never_written_read = set(), set(), set()
    for l in lines[a:b]:
        l = l.split(';')[0].strip()
        if not l or l.endswith(':') or l.startswith('.') or l.startswith(';'): continue
        parts = l.split(None, 1)
        if len(parts) < 2: continue
        op, args = parts
        ops = [x.strip() for x in args.split(',')]
        stores = op.startswith(('global_store', 'scratch_store', 'ds_write', 'buffer_store', 'flat_store')) or op.startswith(('s_', 'v_cmp', 'v_readlane', 'v_readfirstlane'))
        dst = [] if stores else regs(ops[0])
        src = regs(','.join(ops if stores else ops[1:]))
        if op in ('v_fmac_f64_e32', 'v_fmac_f32_e32', 'v_mac_f32_e32', 'v_writelane_b32') or 'fmac' in op: src += dst
        for r in src:
            if r not in written: livein.add(r)
        written.update(dst)
    inv = {r for r in livein if r not in written}
    print(ks[i]); print('  largest loop: %d lines; live-in vector registers %d (never written in the loop: %d, loop-carried: %d)' % (ln, len(livein), len(inv), len(livein) - len(inv)))
