"""Diagnostic (round 5): the all-atoms problem (tests/golden/falc_all.npz) on perturbed columns, HIP against the oracle and against
the oracle's own +-1-ulp-exp runs, iteration by iteration and column by column.  python profiles/diag_all_atoms.py [ncol] [mode]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from lightspinner_amd import fixtures, synth, Engine, _capi
import oracle

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 36
mode = sys.argv[2] if len(sys.argv) > 2 else 'ray-per-lane'
prob, base, raw = fixtures.load_problem_npz(os.path.join(ROOT, 'tests', 'golden', 'falc_all.npz'), phi_compact=False)
blk, (aD, vB, vlos) = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=2468, vlos_sigma=2.0e3)
hip_lib, ora_lib = _capi.load_hip_library(), oracle.load()


def mk(lib, **kw):
    e = Engine(prob, ncol, lib=lib, **kw)
    e.set_columns(0, blk)
    e.set_line_profiles(0, aD, vB, vlos)
    return e


def colerr(a, b):
    a, b = a.reshape(ncol, -1), b.reshape(ncol, -1)
    den = np.maximum(np.abs(b), 1e-300)
    return np.max(np.abs(a - b) / den, axis=1)


hip = mk(hip_lib, sweep_policy=mode)
oras = {}
for u in (0, 1, -1):
    oras[u] = mk(ora_lib)
    ora_lib.dll.lsx_oracle_set_threads(oras[u]._h, 16)
for it in range(1, 9):
    dJ = hip.formal_sol_gamma()
    dJo = {}
    for u in (0, 1, -1):
        ora_lib.dll.lsx_oracle_set_exp_ulp(u)
        dJo[u] = oras[u].formal_sol_gamma()
    eJ = colerr(hip.get(_capi.LSX_J), oras[0].get(_capi.LSX_J))
    eJu = np.maximum(colerr(oras[1].get(_capi.LSX_J), oras[0].get(_capi.LSX_J)), colerr(oras[-1].get(_capi.LSX_J), oras[0].get(_capi.LSX_J)))
    dc, dco = hip.get(_capi.LSX_DJ_COL), oras[0].get(_capi.LSX_DJ_COL)
    w = int(np.argmax(eJ))
    print('it %d dJ %.10g oracle %.10g (+1 %.10g, -1 %.10g)  J err max %.2e col %d (oracle envelope there %.2e, max %.2e)  dJ_col worst rel %.2e' %
          (it, dJ, dJo[0], dJo[1], dJo[-1], eJ.max(), w, eJu[w], eJu.max(), np.max(np.abs(dc / dco - 1))), flush=True)
    if it > 3:
        dP = hip.stat_equil()
        dPo = {}
        for u in (0, 1, -1):
            ora_lib.dll.lsx_oracle_set_exp_ulp(u)
            dPo[u] = oras[u].stat_equil()
        en = colerr(hip.get(_capi.LSX_N), oras[0].get(_capi.LSX_N))
        enu = np.maximum(colerr(oras[1].get(_capi.LSX_N), oras[0].get(_capi.LSX_N)), colerr(oras[-1].get(_capi.LSX_N), oras[0].get(_capi.LSX_N)))
        w = int(np.argmax(en))
        print('     dPops %.10g oracle %.10g (+1 %.10g, -1 %.10g)  n err max %.2e col %d (envelope there %.2e, max %.2e)' %
              (dP, dPo[0], dPo[1], dPo[-1], en.max(), w, enu[w], enu.max()), flush=True)
ora_lib.dll.lsx_oracle_set_exp_ulp(0)
