import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from lightspinner_amd import fixtures, synth, Engine
prob, base, raw = fixtures.load_problem_npz('tests/golden/falc_ca.npz', phi_compact=False)
ncol = 1000
blk, prof = synth.perturbed_columns(prob, base, raw, ncol=ncol)
eng = Engine(prob, ncol)
synth.load_columns(eng, blk, prof)
for _ in range(3):
    eng.formal_sol_gamma_async(); eng.stat_equil_async(); eng.sync()
T = np.zeros(4)
N = 30
for _ in range(N):
    t0 = time.perf_counter(); eng.formal_sol_gamma_async()
    t1 = time.perf_counter(); eng.stat_equil_async()
    t2 = time.perf_counter(); eng.sync()
    t3 = time.perf_counter()
    T += [t1 - t0, t2 - t1, t3 - t2, t3 - t0]
print('per step [ms]: enqueue FS %.3f  enqueue SE %.3f  sync %.3f  total %.3f' % tuple(T / N * 1e3))
