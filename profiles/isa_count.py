#!/usr/bin/env python3
"""Static instruction mix of the sweep kernels' depth loops, from the compiler's assembly:
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -DLSX_WAVES_PER_EU=4 --cuda-device-only -S lsx_sweep.hip -o sweep.s
    python profiles/isa_count.py sweep.s 'ILi2ELi2ELi5ELb0'
Per inner loop (the three phases of a sweep are three loops / peeled bodies): VALU f64, VALU other, transcendental,
DPP / permlane, LDS, vector memory, scalar, s_nop.  The VALU pipe cost estimate is 4 cycles per f64 instruction and
2 per other VALU instruction (wave64 on a SIMD-32; MI355X_MICROARCH.md, cycle constants), 8 per f64 transcendental."""
import re, sys, collections


def classify(op):
    if op.startswith('v_'):
        if 'permlane' in op: return 'xlane'
        if op in ('v_rcp_f64_e32', 'v_rsq_f64_e32', 'v_sqrt_f64_e32') or op.startswith('v_rcp_f64'): return 'trans64'
        if '_f64' in op or 'b64' in op and op.startswith('v_mov_b64'):
            return 'valu64' if not op.startswith('v_mov_b64') else 'mov64'
        return 'valu32'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op == 's_nop': return 'nop'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    return 'other'


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') is False and ':' in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    # inner loops: from "Inner Loop Header" label to the last line tagged "in Loop: Header=<that label>"
    heads = [(i, m.group(1)) for i, l in enumerate(body) if 'Inner Loop Header' in l for m in [re.match(r'\.(LBB\d+_\d+):', l)] if m]
    for hi, name in heads:
        last = hi
        for i in range(hi, len(body)):
            if 'Header=%s ' % name[1:] in body[i] or 'Header=%s\t' % name[1:] in body[i]:
                last = i
        # extend to the next label after `last`
        j = last + 1
        while j < len(body) and not re.match(r'\.LBB\d+_\d+:', body[j]) and not body[j].startswith('; %bb'):
            j += 1
        cnt = collections.Counter()
        dpp = 0
        for l in body[hi:j]:
            l = l.strip()
            if not l or l.startswith((';', '.')):
                continue
            op = l.split()[0]
            c = classify(op)
            cnt[c] += 1
            if '_dpp' in op or 'row_' in l or 'quad_perm' in l:
                dpp += 1
        n = sum(cnt.values())
        if n < 40:
            continue
        pipe = 4 * cnt['valu64'] + 2 * (cnt['valu32'] + cnt['mov64'] * 2 // 2 + cnt['xlane']) + 8 * cnt['trans64']
        print('%-10s lines %5d: valu64 %3d  mov64 %3d  valu32 %3d (dpp %2d)  xlane %2d  trans64 %d  lds %3d  vmem %2d  salu %3d  nop %2d  wait %2d | VALU instr %3d, pipe cycles ~%d'
              % (name, j - hi, cnt['valu64'], cnt['mov64'], cnt['valu32'], dpp, cnt['xlane'], cnt['trans64'], cnt['lds'], cnt['vmem'],
                 cnt['salu'], cnt['nop'], cnt['wait'], cnt['valu64'] + cnt['mov64'] + cnt['valu32'] + cnt['xlane'] + cnt['trans64'], pipe))


if __name__ == '__main__':
    main()
