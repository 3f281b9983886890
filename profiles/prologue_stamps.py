#!/usr/bin/env python3
"""Where a ray-serial sweep wave's PROLOGUE goes (a -DLSX_CLOCK build of lsx_sweep_rs.hip: bash profiles/mkvariant.sh clock -DLSX_CLOCK rs):
    python3 profiles/prologue_stamps.py ab_so/clock.so [c3|c4] [ncol]
shader-clock stamps at the wave's first instruction, behind the first workgroup barrier (tile tables, operand ring, folded
cross-sections in LDS), behind the second (quadrature constants), behind the per-slot lane constants, behind the boundary condition
(= first depth step), and around the depth loop -- medians over the down-going waves of the column groups 5, 105, ... per tile class."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lightspinner_amd import fixtures, synth, Engine, _capi, drivers
so = os.path.abspath(sys.argv[1]); wl = sys.argv[2] if len(sys.argv) > 2 else 'c4'
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
cls = {}
for r in D:
    if r[7] == 0 or r[11] % 100 != 5:
        continue
    cls.setdefault((int(r[9]), int(r[12]), int(r[13]), int(r[14])), []).append(r)
print('%s, %d columns, library %s: medians in microseconds at the clock the wave itself measured' % (wl, ncol, os.path.basename(so)))
for key, rows in sorted(cls.items()):
    R = np.array(rows)
    ghz = np.median(R[:, 7] / R[:, 15]) * 0.1
    us = lambda c: np.median(R[:, c]) / ghz / 1e3
    print('class (%d,%d,%d,%d) %3d waves, %.2f GHz: barrier 1 at %5.1f  barrier 2 at %5.1f  lane constants at %5.1f  streams of depths 0, 1 requested at %5.1f  first step at %5.1f | depth loop %6.1f us'
          % (key + (len(R), ghz, us(2), us(3), us(4), us(5), us(0), np.median(R[:, 15]) / 100.0)))
eng.close()
