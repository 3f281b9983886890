import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from lightspinner_amd import fixtures, synth, Engine, _capi
res = {}
for name in ('falc_ca.npz', 'falc_cah.npz'):
    prob, base, raw = fixtures.load_problem_npz('tests/golden/' + name, phi_compact=False)
    blk = synth.perturbed_columns(prob, base, raw, ncol=40)
    outs = []
    for lds in (False, True):
        if lds: os.environ['LSX_SE_LDS'] = '1'
        else: os.environ.pop('LSX_SE_LDS', None)
        e = Engine(prob, 40); e.set_columns(0, blk)
        for i in range(6):
            e.formal_sol_gamma(); dp = e.stat_equil()
        outs.append((e.get(_capi.LSX_N), dp, e.get(_capi.LSX_DPOPS_COL)))
    print(name, 'bitwise n:', np.array_equal(outs[0][0], outs[1][0]), 'dP', outs[0][1] == outs[1][1], np.array_equal(outs[0][2], outs[1][2]))
