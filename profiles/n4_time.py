#!/usr/bin/env python3
"""time of one formal solution with the linear and with the parabolic rule (N4): python3 profiles/n4_time.py [c3|c4] [ncol]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lightspinner_amd import fixtures, synth, Engine
wl = sys.argv[1] if len(sys.argv) > 1 else 'c3'
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else (1000 if wl == 'c3' else 1250)
prob, base, raw = fixtures.load_problem_npz(os.path.join(ROOT, 'tests', 'golden', 'falc_cah.npz' if wl == 'c4' else 'falc_ca.npz'), phi_compact=False)
blk, prof = synth.perturbed_columns(prob, base, raw, ncol=ncol)
eng = Engine(prob, ncol)
synth.load_columns(eng, blk, prof)
# (the parabolic rule on the ray-serial kernels where a class has an instance there, and -- 'parabolic/lane' -- on one ray per lane only)
for rule, policy in (('linear', 'auto'), ('parabolic', 'auto'), ('parabolic', 'ray-per-lane'), ('linear', 'auto')):
    synth.load_columns(eng, blk, prof)        # every configuration from the same starting populations
    eng.set_formal_solver(rule)
    eng.set_sweep_policy(policy)
    for _ in range(3):
        eng.formal_sol_gamma(); eng.stat_equil()
    t, s = eng.time_formal_sol(2, 10)
    print('%s %-9s %-12s ncol=%d  formal solution %.3f ms' % (wl, rule, policy, ncol, t))
