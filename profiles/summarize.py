#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by profiles/collect.sh into one JSON (stdout).

usage: python profiles/summarize.py gpurun_out/<tag> [ncol]

The sweep of one formal-solution call is several kernels (`lsx_sweep_kernel<slots, lines, rays, sca>`, one per
tile class) launched side by side on forked streams, so the figure that corresponds to bench.py's
`roofline.avg_launch_ms` is the SPAN of those launches (first start -> last end), not one kernel's duration;
HBM bytes are summed over the classes of a call.  FETCH_SIZE is in KiB and counts half the bytes on gfx950
(MI355X_MICROARCH.md; factor checked by the calibration pass): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv, glob, json, os, re, sys, collections


def short(name):
    m = re.search(r'(lsx_sweep_(?:rs_)?kernel(?:_all)?<[^>]*>)', name)
    if m:
        return m.group(1).replace(' ', '')
    m = re.search(r'(k_\w+|__amd_\w+)', name)
    return m.group(1) if m else name[:40]


def main():
    tag = sys.argv[1]
    ncol = int(sys.argv[2]) if len(sys.argv) > 2 else None
    out = {'tag': os.path.basename(tag)}
    kt = glob.glob(tag + '_kt/*kernel_trace.csv')
    if kt:
        rows = [r for r in csv.DictReader(open(kt[0]))]
        per = collections.defaultdict(list)
        for r in rows:
            per[short(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        out['kernels_avg_us'] = {k: {'calls': len(v), 'avg_us': round(sum(v) / len(v) / 1e3, 2)} for k, v in sorted(per.items())}
        sw = sorted([r for r in rows if 'lsx_sweep_' in r['Kernel_Name']], key=lambda r: int(r['Start_Timestamp']))
        classes = sorted({short(r['Kernel_Name']) for r in sw})
        n = len(classes)
        spans = []
        for i in range(0, len(sw) - n + 1, n):
            grp = sw[i:i + n]
            if len({short(r['Kernel_Name']) for r in grp}) != n:
                continue        # a call whose launches interleave with the next one is skipped
            spans.append(max(int(r['End_Timestamp']) for r in grp) - min(int(r['Start_Timestamp']) for r in grp))
        if spans:
            out['sweep'] = {'classes': classes, 'calls': len(spans), 'span_avg_ms': round(sum(spans) / len(spans) / 1e6, 4),
                            'span_min_ms': round(min(spans) / 1e6, 4), 'span_max_ms': round(max(spans) / 1e6, 4)}
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(tag + '_pmc*')):
        for f in glob.glob(os.path.join(d, '*counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                if 'lsx_sweep_' in r['Kernel_Name']:
                    ctr[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    # ---- whole-call HBM traffic (round 5): every kernel of a formal solution -- the sweeps, the fast-continuum kernels around them,
    # the operand-table build, the Gamma epilogue -- summed over a pass and divided by the number of calls in it (= launches of
    # k_gamma_finish).  SURVEY 8d: re-reads caused by kernel splitting do not count as algorithmic bytes, so they have to be visible.
    FS_KERNELS = ('lsx_sweep_', 'k_fast_', 'k_gamma_finish', 'k_build_optab')
    allk = collections.defaultdict(lambda: collections.defaultdict(float))
    ncalls = collections.Counter()
    for d in sorted(glob.glob(tag + '_pmc*')):
        for f in glob.glob(os.path.join(d, '*counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] not in ('FETCH_SIZE', 'WRITE_SIZE'):
                    continue
                nm = short(r['Kernel_Name'])
                if any(x in r['Kernel_Name'] for x in FS_KERNELS):
                    allk[r['Counter_Name']][nm] += float(r['Counter_Value'])
                    if 'k_gamma_finish' in r['Kernel_Name']:
                        ncalls[r['Counter_Name']] += 1
    # ---- ceiling model (round 6): per kernel of a call, what the bytes it really moves and the vector instructions it really issues
    # allow at best -- t_min = max(counter bytes / ACHIEVABLE_BW, VALU instructions * 4 cycles / (1024 SIMDs * held clock)) -- against
    # what it takes alone (the --pmc passes serialise the kernels of a call: End - Start of a dispatch in the lightest pass, FETCH_SIZE +
    # GRBM_GUI_ACTIVE, is the kernel alone on the machine).  ACHIEVABLE_BW: the 6.29 TB/s MI355X_MICROARCH.md measures; held clock: what
    # the chip holds under the fp64 sweeps (in-kernel s_memtime / s_memrealtime, profiles/r03_bound_evidence.md 7: 1.9 GHz C3, 2.1 C4).
    valu_all = collections.defaultdict(float)
    dur_all = collections.defaultdict(list)
    nvalu_calls = 0
    for d in sorted(glob.glob(tag + '_pmc*')):
        for f in glob.glob(os.path.join(d, '*counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                if not any(x in r['Kernel_Name'] for x in FS_KERNELS):
                    continue
                nm = short(r['Kernel_Name'])
                if r['Counter_Name'] == 'SQ_INSTS_VALU':
                    valu_all[nm] += float(r['Counter_Value'])
                    if 'k_gamma_finish' in r['Kernel_Name']:
                        nvalu_calls += 1
                if r['Counter_Name'] == 'FETCH_SIZE' and d.endswith('pmc3'):
                    dur_all[nm].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    if allk.get('FETCH_SIZE') and allk.get('WRITE_SIZE') and ncalls['FETCH_SIZE'] and ncalls['WRITE_SIZE']:
        per_kernel = {}
        for nm in sorted(set(allk['FETCH_SIZE']) | set(allk['WRITE_SIZE'])):
            per_kernel[nm] = (2.0 * allk['FETCH_SIZE'].get(nm, 0.0) / ncalls['FETCH_SIZE'] + allk['WRITE_SIZE'].get(nm, 0.0) / ncalls['WRITE_SIZE']) * 1024.0
        out['fs_call_hbm_bytes_per_call_by_kernel'] = {k: round(v) for k, v in per_kernel.items()}
        out['fs_call_hbm_bytes_per_call'] = sum(per_kernel.values())
        out['fs_calls_counted'] = dict(ncalls)
        if ncol:
            out['fs_call_hbm_bytes_per_call_per_column'] = out['fs_call_hbm_bytes_per_call'] / ncol
        if nvalu_calls and dur_all:
            wl = sys.argv[3] if len(sys.argv) > 3 else 'c3'
            held = {'c3': 1.9e9, 'c4': 2.1e9}.get(wl, 2.0e9)
            bw = 6.29e12
            cm = {}
            calls3 = max(len(dur_all.get(next((k for k in dur_all if 'k_gamma_finish' in k), ''), [])), 1)
            for nm in sorted(per_kernel):
                v = valu_all.get(nm, 0.0) / nvalu_calls
                alone = sum(dur_all.get(nm, [])) / calls3 * 1e-9            # seconds per call (a kernel launched several times per call: summed)
                tb, tv = per_kernel[nm] / bw, v * 4.0 / (1024.0 * held)
                cm[nm] = dict(hbm_bytes=round(per_kernel[nm]), valu_insts=round(v), alone_ms=round(alone * 1e3, 4), t_bytes_ms=round(tb * 1e3, 4),
                              t_valu_ms=round(tv * 1e3, 4), bound='hbm' if tb >= tv else 'valu',
                              frac_of_ceiling=round(max(tb, tv) / alone, 4) if alone else None)
            tb = sum(per_kernel.values()) / bw
            tv = sum(valu_all.values()) / nvalu_calls * 4.0 / (1024.0 * held)
            sw = [k for k in cm if 'lsx_sweep_' in k]
            out['ceiling_model'] = dict(
                achievable_bw_Bps=bw, held_clock_Hz=held, per_kernel=cm,
                call=dict(hbm_bytes=round(sum(per_kernel.values())), valu_insts=round(sum(valu_all.values()) / nvalu_calls),
                          t_bytes_ms=round(tb * 1e3, 4), t_valu_ms=round(tv * 1e3, 4), bound='hbm' if tb >= tv else 'valu',
                          alone_sum_ms=round(sum(c['alone_ms'] for c in cm.values()), 4)),
                sweeps=dict(hbm_bytes=round(sum(cm[k]['hbm_bytes'] for k in sw)), valu_insts=round(sum(cm[k]['valu_insts'] for k in sw)),
                            t_bytes_ms=round(sum(cm[k]['hbm_bytes'] for k in sw) / bw * 1e3, 4),
                            t_valu_ms=round(sum(cm[k]['valu_insts'] for k in sw) * 4.0 / (1024.0 * held) * 1e3, 4)),
                note='t_min = max(t_bytes, t_valu); a kernel alone cannot be faster than its own t_min, a call cannot be faster than the call\'s '
                     '(its kernels overlap: the call is work conserving); bench.py divides the call\'s t_min by the LIVE duration')
    if ctr:
        pc = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in ctr.items()}
        out['sweep_counters_per_launch'] = {k: {c: round(v, 1) for c, v in d.items()} for k, d in sorted(pc.items())}
        tot = collections.Counter()
        for d in pc.values():
            for c, v in d.items():
                tot[c] += v
        if 'FETCH_SIZE' in tot and 'WRITE_SIZE' in tot:
            hbm = (2.0 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024.0
            out['sweep_hbm_bytes_per_call'] = hbm
            if ncol:
                out['sweep_hbm_bytes_per_call_per_column'] = hbm / ncol
        if 'SQ_WAVES' in tot:
            out['sweep_per_wave'] = {k: {c.replace('SQ_INSTS_', '').replace('SQ_', ''): round(d[c] / d['SQ_WAVES'], 1)
                                         for c in d if c != 'SQ_WAVES'} for k, d in sorted(pc.items()) if 'SQ_WAVES' in d}
    if ctr:
        # ---- what binds the sweep: VALU issue against memory, from the counters alone (per class; the PMC passes serialise
        # the classes, so each ratio is that class alone on the machine)
        #   VALU busy   = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * cycles)        (quad-cycles, MI355X_MICROARCH.md; = rocprofv3's VALUBusy)
        #   cycles      = GRBM_GUI_ACTIVE / 8                               (the counter is summed over the 8 XCDs)
        #   f64 share   = (FMA + ADD + MUL + TRANS)_F64 / SQ_INSTS_VALU
        nsimd = 1024.0
        bind = {}
        for k, d in sorted(pc.items()):
            b = {}
            cyc = d.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
            if cyc and 'SQ_ACTIVE_INST_VALU' in d:
                b['kernel_cycles'] = round(cyc)
                b['valu_busy_frac'] = round(d['SQ_ACTIVE_INST_VALU'] * 4.0 / (nsimd * cyc), 4)
                for nm in ('SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_ANY'):
                    if nm in d:
                        b[nm[3:].lower() + '_frac_of_simd_cycles'] = round(d[nm] * 4.0 / (nsimd * cyc), 4)
            if 'SQ_WAVE_CYCLES' in d:
                for nm in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU'):
                    if nm in d:
                        b[nm[3:].lower() + '_share_of_wave_cycles'] = round(d[nm] / d['SQ_WAVE_CYCLES'], 4)
            f64 = sum(d.get('SQ_INSTS_VALU_%s_F64' % x, 0.0) for x in ('FMA', 'ADD', 'MUL', 'TRANS'))
            if f64 and d.get('SQ_INSTS_VALU'):
                b['f64_insts'] = round(f64)
                b['f64_share_of_valu_insts'] = round(f64 / d['SQ_INSTS_VALU'], 4)
                b['valu_pipe_cycles_est'] = round(4.0 * f64 + 2.0 * (d['SQ_INSTS_VALU'] - f64))
                if cyc:
                    b['valu_pipe_frac_est'] = round(b['valu_pipe_cycles_est'] / (nsimd * cyc), 4)
            if b:
                bind[k] = b
        if bind:
            out['sweep_binding_resource'] = bind
        fig = {}
        if 'sweep_hbm_bytes_per_call_per_column' in out:
            fig['hbm_bytes_per_call_per_column'] = out['sweep_hbm_bytes_per_call_per_column']
        if ncol and 'SQ_INSTS_VALU' in tot:
            fig['valu_insts_per_call_per_column'] = tot['SQ_INSTS_VALU'] / ncol
            f64 = sum(tot.get('SQ_INSTS_VALU_%s_F64' % x, 0.0) for x in ('FMA', 'ADD', 'MUL', 'TRANS'))
            if f64:
                fig['f64_share_of_valu_insts'] = f64 / tot['SQ_INSTS_VALU']
        # (round 2 quoted SQ_ACTIVE_INST_VALU * 4 / (SIMDs x GRBM cycles) as "VALU busy"; its sibling ratio for all instructions
        # exceeds 1, so it is not a utilisation and is no longer published; the pipe estimate per class stays in
        # sweep_binding_resource, the clock the chip actually holds is measured in the kernel: profiles/stamps.py)
        if 'fs_call_hbm_bytes_per_call_per_column' in out:
            fig['hbm_bytes_per_call_per_column_all_kernels'] = out['fs_call_hbm_bytes_per_call_per_column']
        if 'ceiling_model' in out and ncol:
            c = out['ceiling_model']
            fig['valu_insts_per_call_per_column_all_kernels'] = c['call']['valu_insts'] / ncol
            fig['held_clock_Hz'] = c['held_clock_Hz']
            fig['achievable_bw_Bps'] = c['achievable_bw_Bps']
            fig['ceiling_per_kernel'] = {k: dict(v, hbm_bytes_per_column=v['hbm_bytes'] / ncol, valu_insts_per_column=v['valu_insts'] / ncol)
                                         for k, v in c['per_kernel'].items()}
            fig['ceiling_profiled_columns'] = ncol
        if fig:
            out['figures'] = fig
    cal = glob.glob(tag + '_calib/*counter_collection.csv')
    log = tag + '_calib.log'
    if cal and os.path.exists(log):
        fetch = [float(r['Counter_Value']) for r in csv.DictReader(open(cal[0])) if 'k_calib_read' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
        byts = [float(x) for x in re.findall(r'bytes_read=(\d+)', open(log).read())]
        if fetch and len(fetch) == len(byts):
            out['fetch_size_calibration_factor'] = [round(b / (f * 1024.0), 4) for b, f in zip(byts, fetch)]
    print(json.dumps(out, indent=1))
    # `python profiles/summarize.py gpurun_out/<tag> <ncol> <workload>` also records the per-column figures bench.py quotes
    if len(sys.argv) > 3 and out.get('figures'):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pmc_figures.json')
        try:
            allfig = json.load(open(path))
        except Exception:
            allfig = {}
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import srchash
        allfig[sys.argv[3]] = dict(out['figures'], csrc_hash=srchash.csrc_hash(),
                                   source='rocprofv3 --pmc passes of `bench.py --workload %s` (profiles/collect.sh %s), '
                                   'summed over the sweep classes of a call' % (sys.argv[3], os.path.basename(tag)))
        json.dump(allfig, open(path, 'w'), indent=1)


if __name__ == '__main__':
    main()
