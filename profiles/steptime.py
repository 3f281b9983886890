#!/usr/bin/env python3
"""Host-side split of a MALI step and a plain loop for kernel traces:
    python3 profiles/steptime.py [c3|c4] [ncol] [steps]
(with LSX_SERIAL=1 every class runs on the context's stream one after the other: per-kernel durations in a rocprofv3
kernel trace are then what each kernel costs alone; profiles/kt_sum.py <trace.csv> <steps + 3> tabulates them)"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from lightspinner_amd import fixtures, synth, Engine
wl = sys.argv[1] if len(sys.argv) > 1 else 'c3'
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else (1250 if wl == 'c4' else 1000)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 30
prob, base, raw = fixtures.load_problem_npz('tests/golden/falc_cah.npz' if wl == 'c4' else 'tests/golden/falc_ca.npz', phi_compact=False)
blk, prof = synth.perturbed_columns(prob, base, raw, ncol=ncol)
eng = Engine(prob, ncol)
synth.load_columns(eng, blk, prof)
if os.environ.get('LSX_STEP_SOLVER'):      # 'parabolic' (N4); with LSX_FS_ONLY=1: the strongly perturbed ensemble does not converge under that rule
    eng.set_formal_solver(os.environ['LSX_STEP_SOLVER'])
if os.environ.get('LSX_STEP_POLICY'):
    eng.set_sweep_policy(os.environ['LSX_STEP_POLICY'])
fs_only = os.environ.get('LSX_FS_ONLY') == '1'      # ablation variants (meaningless results): formal solutions only
se = (lambda: None) if fs_only else eng.stat_equil_async
for _ in range(3):
    eng.formal_sol_gamma_async(); se(); eng.sync()
T = np.zeros(4)
for _ in range(N):
    t0 = time.perf_counter(); eng.formal_sol_gamma_async()
    t1 = time.perf_counter(); se()
    t2 = time.perf_counter(); eng.sync()
    t3 = time.perf_counter()
    T += [t1 - t0, t2 - t1, t3 - t2, t3 - t0]
print('%s ncol=%d per step [ms]: enqueue FS %.3f  enqueue SE %.3f  sync %.3f  total %.3f' % ((wl, ncol) + tuple(T / N * 1e3)))
