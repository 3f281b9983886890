import sys, os, numpy as np
ROOT='/root/repo' if os.path.exists('/root/repo/tests') else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
from conftest import golden
from lightspinner_amd import fixtures, synth, Engine, _capi
import oracle
ora_lib = oracle.load(); hip = _capi.load_hip_library()
for name in ('falc_ca.npz',):
    prob, base, raw = fixtures.load_problem_npz(golden(name), phi_compact=False)
    for ncol in (10, 9):
        blk, (aD, vB, vlos) = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=1234, vlos_sigma=2.0e3)
        o = Engine(prob, ncol, lib=ora_lib); o.set_columns(0, blk); o.set_line_profiles(0, aD, vB, vlos); o.formal_sol_gamma()
        Jo = o.get(_capi.LSX_J)
        for opt in ('fold=1', 'fold=0'):
            e = Engine(prob, ncol, lib=hip, sweep_policy='ray-serial', options=opt); e.set_columns(0, blk); e.set_line_profiles(0, aD, vB, vlos); e.formal_sol_gamma()
            J = e.get(_capi.LSX_J)
            err = np.abs(J - Jo) / np.abs(Jo)
            print(name, ncol, opt, 'max', err.max(), 'classes', e.effective_options().split('classes=')[1])
            bad = np.argwhere(err > 1e-10)
            if len(bad):
                print('  bad cols', sorted(set(bad[:,0])), 'la range', bad[:,1].min(), bad[:,1].max(), 'k range', bad[:,2].min(), bad[:,2].max(), 'count', len(bad))
                la_err = err.max(axis=(0,2)); print('  worst la', np.argsort(la_err)[-8:], la_err[np.argsort(la_err)[-8:]])
                k_err = err.max(axis=(0,1)); print('  err by k (first 10, last 10)', k_err[:10], k_err[-10:])
            e.close()
        o.close()
