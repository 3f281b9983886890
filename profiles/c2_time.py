#!/usr/bin/env python3
"""the single-column case (C2) timed a few times in one process: python3 profiles/c2_time.py [repeats]
(first run = cold: code objects, first launches; the later ones are what a caller iterating many columns one by one sees)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lightspinner_amd import fixtures, Engine, drivers
p1, b1, r1 = fixtures.load_problem_npz(os.path.join(ROOT, 'tests', 'golden', 'falc_ca.npz'))
e1 = Engine(p1, 1)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    e1.set_columns(0, b1)
    t0 = time.perf_counter()
    h = drivers.iterate_mali_engine(e1)
    dt = time.perf_counter() - t0
    print('run %d: iterations %d  %.3f ms  %.1f us per iteration' % (r, h.n_iter, dt * 1e3, dt / h.n_iter * 1e6))
