#!/usr/bin/env python3
"""Where the wall time of the CaII response function (C5 / N2, lightspinner_amd/response.py) goes:  python3 profiles/c5_breakdown.py
cProfile of one run after a warm-up run, cumulative times of the driver's own steps."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lightspinner_amd import fixtures, response, _capi
prob, base, raw = fixtures.load_problem_npz(os.path.join(ROOT, 'tests', 'golden', 'falc_ca.npz'))
fx = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'rf_ca_inputs.npz')))
ks = list(range(int(fx['Nspace'])))
lib = _capi.load_hip_library()
response.run_response_function(prob, base, fx, ks[:2], lib=lib)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
out = response.run_response_function(prob, base, fx, ks, lib=lib)
pr.disable()
print('wall %.1f ms, %d batch iterations (max over columns), base column %d' % ((time.perf_counter() - t0) * 1e3, int(out['n_iter'].max()), out['n_iter_base']))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print(s.getvalue()[:6000])
