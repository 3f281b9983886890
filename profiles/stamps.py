#!/usr/bin/env python3
"""Per-segment shader-clock cycles of one depth step of the sweep, from a -DLSX_STAMPS build of the library
(bash profiles/mkvariant.sh stamps -DLSX_STAMPS; the in-tree library never executes a stamp):
    python3 profiles/stamps.py ab_so/stamps.so [c3|c4] [ncol]
The machine is loaded with the full workload; the records are the down-going waves of every tile of column 5.  A stamp
waits for the wave's LDS operations only (s_memtime needs lgkmcnt(0)), not for its vector-memory loads: time a segment
spends in `s_waitcnt vmcnt` for the streams it consumes stays in that segment.  Read the SHARES, not the total: the
stamps themselves cost about 40 cycles each and forbid overlap across segment borders (cdna_hip_programming.md, stamps)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lightspinner_amd import fixtures, synth, Engine, _capi, drivers

so = os.path.abspath(sys.argv[1])
wl = sys.argv[2] if len(sys.argv) > 2 else 'c3'
ncol = int(sys.argv[3]) if len(sys.argv) > 3 else (1000 if wl == 'c3' else 1250)
prob, base, raw = fixtures.load_problem_npz(os.path.join(ROOT, 'tests', 'golden', 'falc_cah.npz' if wl == 'c4' else 'falc_ca.npz'), phi_compact=False)
blk, prof = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=1234, vlos_sigma=2.0e3)
_capi._share_hip_runtime_with_torch()
lib = _capi.LsxLibrary(so)
eng = Engine(prob, ncol, lib=lib)
synth.load_columns(eng, blk, prof)
for _ in range(6):
    drivers.mali_step(eng)
buf = (C.c_ulonglong * (1024 * 16))()
lib.dll.lsx_hip_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
lib.check(lib.dll.lsx_hip_debug_read(eng._h, buf))
D = np.array(buf[:], dtype=np.uint64).reshape(1024, 16).astype(np.float64)
names = ['top: stream prefetch issue, operand table reads', 'stamp overhead (two stamps back to back)', 'pass 1: chi, eta of the per-ray slots',
         'source function + formal solution (2 rcp, exp)', 'angle sums through LDS (J, Psibar, Psi* phi)', 'pass 2: Gamma integrands + lane reduction',
         'J store / half-J add, dJ', 'after the loop']
classes = {}
for r in D:
    if r[:8].sum() == 0 or r[11] % 100 != 5:
        continue
    key = (int(r[9]), int(r[12]), int(r[13]), int(r[14]))
    classes.setdefault(key, []).append(np.concatenate([r[:8], r[15:16]]))
out = {}
Ns = prob.Nspace
for key, rows in sorted(classes.items()):
    rows = np.array(rows)
    clock_ghz = float(np.median(rows[:, 7] / rows[:, 8])) * 0.1      # shader ticks per 100 MHz tick
    wave_us = float(np.median(rows[:, 8])) / 100.0
    m = np.mean(rows[:, :8], axis=0) / Ns
    ov = m[1]
    seg = np.maximum(m - ov, 0.0)
    seg[1] = 0.0
    tot = max(seg[:7].sum(), 1e-9)
    print('class per-ray slots %d, lines %d, linked %d, topo %d  (%d tiles sampled): %.0f cycles per depth step net of stamps (stamp %.0f)'
          % (key + (len(rows), tot, ov)))
    print('    in-kernel clock %.2f GHz (s_memtime / s_memrealtime), the wave lived %.1f us' % (clock_ghz, wave_us))
    if rows[:, 2:7].sum() == 0:       # -DLSX_CLOCK build: T[0] = prologue ticks, T[1] = loop start (100 MHz ticks)
        print('    prologue %.1f us (median; entry -> first depth step), loop %.1f us' % (float(np.median(rows[:, 0])) / clock_ghz / 1e3, wave_us))
        st = np.sort(rows[:, 1]) / 100.0
        print('    loop starts of the %d sampled waves relative to the first [us]: %s' % (len(st), ' '.join('%.0f' % x for x in (st - st[0])[::max(1, len(st) // 16)])))
        print('    loop lifetimes [us] min %.1f median %.1f max %.1f' % (rows[:, 8].min() / 100, np.median(rows[:, 8]) / 100, rows[:, 8].max() / 100))
    for i in (0, 2, 3, 4, 5, 6):
        print('    %-52s %7.0f  %5.1f %%' % (names[i], seg[i], 100 * seg[i] / tot))
    out['npt%d_nl%d_lk%d_topo%d' % key] = dict(tiles=len(rows), clock_ghz=clock_ghz, wave_lifetime_us=wave_us, cycles_per_step=tot, stamp_cycles=ov,
                                              segments={names[i]: dict(cycles=float(seg[i]), share=float(seg[i] / tot)) for i in (0, 2, 3, 4, 5, 6)})
json.dump(dict(workload=wl, ncol=ncol, library=os.path.basename(so), classes=out), open(os.path.join(ROOT, 'gpurun_out', 'stamps_%s.json' % wl), 'w'), indent=1)
eng.close()
