#!/usr/bin/env python3
"""HIP-event time of one formal solution of the single FALC column (C2): python3 profiles/c2_kernel_time.py  -> ms total / sweep"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lightspinner_amd import fixtures, Engine, drivers
p1, b1, r1 = fixtures.load_problem_npz(os.path.join(ROOT, 'tests', 'golden', 'falc_ca.npz'))
e1 = Engine(p1, 1)
e1.set_columns(0, b1)
for _ in range(3):
    drivers.mali_step(e1)
best = min(e1.time_formal_sol(5, 50) for _ in range(3))
print('%s: fs %.1f us  sweep part %.1f us' % (os.environ.get('TAG', ''), best[0] * 1e3, best[1] * 1e3))
