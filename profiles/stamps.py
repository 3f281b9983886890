#!/usr/bin/env python3
"""Diagnostic: build the sweep kernel with -DLSX_STAMPS, run one FS call and print where a depth
step spends its shader cycles (shares only -- the stamped build is slower than the real one)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 1
subprocess.check_call('make -s -C %s/lightspinner_amd/csrc clean && make -s -j8 -C %s/lightspinner_amd/csrc HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DLSX_STAMPS -DLSX_WAVES_PER_EU=4"' % (ROOT, ROOT), shell=True)
from lightspinner_amd import fixtures, synth, Engine, _capi
prob, block, raw = fixtures.load_problem_npz(os.path.join(ROOT, 'tests/golden/falc_ca.npz'), phi_compact=False)
batch = synth.perturbed_columns(prob, block, raw, ncol=ncol) if ncol > 1 else block
lib = _capi.load_hip_library()
eng = Engine(prob, ncol, lib=lib); eng.set_columns(0, batch)
for _ in range(3): eng.formal_sol_gamma()
buf = (C.c_ulonglong * (64 * 16))()
lib.dll.lsx_hip_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
lib.check(lib.dll.lsx_hip_debug_read(eng._h, buf))
a = np.array(buf[:], dtype=np.uint64).reshape(64, 16)
names = ['loads/top', 'fast pre', 'pass1', 'sweep', 'angle sums', 'pass2', 'fast post', 'J/end']
for r in a:
    if r[:8].sum() == 0: continue
    tot = float(r[:8].sum())
    print('tile %d nP %d nF %d col %d: cycles/step %.0f | ' % (r[8], r[9], r[10], r[11], tot / prob.Nspace) + ' '.join('%s %.0f%%' % (n, 100 * v / tot) for n, v in zip(names, r[:8])))
