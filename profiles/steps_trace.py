#!/usr/bin/env python3
"""C3 columns stepped with drivers.mali_steps (pipelined) or drivers.mali_step (plain): python3 profiles/steps_trace.py [plain|pipe] [ncol] [nsteps]
prints the host-clock step times; run under rocprofv3 --kernel-trace for the timeline (profiles/timeline.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lightspinner_amd import fixtures, synth, Engine, drivers
mode = sys.argv[1] if len(sys.argv) > 1 else 'pipe'
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
prob, base, raw = fixtures.load_problem_npz(os.path.join(ROOT, 'tests', 'golden', 'falc_ca.npz'), phi_compact=False)
blk, prof = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=1234, vlos_sigma=2.0e3)
eng = Engine(prob, ncol)
synth.load_columns(eng, blk, prof)
for _ in range(5):
    drivers.mali_step(eng)
per = []
t0 = ta = time.perf_counter()
if mode == 'plain':
    for _ in range(nsteps):
        drivers.mali_step(eng)
        tb = time.perf_counter(); per.append(tb - ta); ta = tb
else:
    for _ in drivers.mali_steps(eng, nsteps):
        tb = time.perf_counter(); per.append(tb - ta); ta = tb
eng.sync()
tot = time.perf_counter() - t0
per = np.array(per) * 1e3
print('%s ncol=%d: %.4f ms per step (total / n), per-step median %.4f min %.4f max %.4f' % (mode, ncol, tot / nsteps * 1e3, np.median(per), per.min(), per.max()))
