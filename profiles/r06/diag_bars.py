#!/usr/bin/env python3
"""Where the HIP library's J deviates most from the oracle's after 8 MALI iterations of the C4-shaped test columns, and what the
oracle's own +-1-ulp-exp runs do there (tests/envelope.py SequenceBars): python3 profiles/r06/diag_bars.py [fixture] [ncol] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle, envelope
from lightspinner_amd import fixtures, synth, Engine, _capi
from conftest import golden
name = sys.argv[1] if len(sys.argv) > 1 else 'falc_cah.npz'
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 41
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 4321
iters = 8
ora = oracle.load(); hip = _capi.load_hip_library()
prob, base, raw = fixtures.load_problem_npz(golden(name), phi_compact=False)
blk, (aD, vB, vlos) = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=seed, vlos_sigma=2.0e3)
def make(lib=ora):
    e = Engine(prob, ncol, lib=lib); e.set_columns(0, blk); e.set_line_profiles(0, aD, vB, vlos)
    if lib is ora: ora.dll.lsx_oracle_set_threads(e._h, 16)
    return e
bars = envelope.SequenceBars(ora, make, prob, iters, 3)
h, o = make(hip), make()
for it in range(1, iters + 1):
    h.formal_sol_gamma(); o.formal_sol_gamma()
    J, Jo = h.get(_capi.LSX_J), o.get(_capi.LSX_J)
    rel = np.abs(J - Jo) / np.maximum(np.abs(Jo), 1e-300)
    c, la, k = np.unravel_index(np.argmax(rel), rel.shape)
    sp = np.abs(bars.runs[1][it - 1][_capi.LSX_J] - bars.runs[-1][it - 1][_capi.LSX_J]) / np.maximum(np.abs(bars.runs[0][it - 1][_capi.LSX_J]), 1e-300)
    print('it %d: max rel dJ %.2e at col %d la %d (%.3f nm) k %d: J %.6e, oracle spread there %.2e, max spread anywhere %.2e; 99.9 pct rel dev %.2e; entries above 3e-10: %d'
          % (it, rel.max(), c, la, prob.wavelength[la], k, Jo[c, la, k], sp[c, la, k], sp.max(), np.quantile(rel, 0.999), int((rel > 3e-10).sum())))
    if it > 3:
        h.stat_equil(); o.stat_equil()
        n, no = h.get(_capi.LSX_N), o.get(_capi.LSX_N)
        rn = np.abs(n - no) / np.abs(no)
        cc, l, kk = np.unravel_index(np.argmax(rn), rn.shape)
        print('      n: max rel %.2e at col %d level %d k %d; bars %s' % (rn.max(), cc, l, kk, ['%.1e' % b for b in bars.n_bar(it - 1)]))
