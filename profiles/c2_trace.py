#!/usr/bin/env python3
"""kernel trace helper for the single-column case (C2): python3 profiles/c2_trace.py  (run under rocprofv3 --kernel-trace)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lightspinner_amd import fixtures, Engine, drivers
p1, b1, r1 = fixtures.load_problem_npz(os.path.join(ROOT, 'tests', 'golden', 'falc_ca.npz'))
e1 = Engine(p1, 1)
e1.set_columns(0, b1)
t0 = time.perf_counter()
h = drivers.iterate_mali_engine(e1)
print('iterations', h.n_iter, 'seconds', time.perf_counter() - t0)
