#!/usr/bin/env python3
"""One variant of the HIP library, timed and checked on one box:
    python tests/ab_run.py ab_so/x.so [c3|c4] [ncol]
prints ONE line: sweep / formal-solution / MALI-step times (HIP events in the library, host clock for the step) and the
parity of the variant against the oracle on 40 columns of the same ensemble, on the mapping the timed context ran (first call and after 6 iterations).
Used by profiles/ab.sh to compare prebuilt variants interleaved on the same GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from lightspinner_amd import fixtures, synth, Engine, _capi, drivers
from conftest import relerr, gamma_err

so = os.path.abspath(sys.argv[1])
wl = sys.argv[2] if len(sys.argv) > 2 else 'c3'
ncol = int(sys.argv[3]) if len(sys.argv) > 3 else (1000 if wl == 'c3' else 1250)
fixture = os.path.join(ROOT, 'tests', 'golden', 'falc_cah.npz' if wl == 'c4' else 'falc_ca.npz')
prob, base, raw = fixtures.load_problem_npz(fixture, phi_compact=False)
blk, prof = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=1234, vlos_sigma=2.0e3)
_capi._share_hip_runtime_with_torch()
lib = _capi.LsxLibrary(so)


def load(eng, b, p, n):
    synth.load_columns(eng, b.slice(0, n), tuple(x[:n] for x in p))


solver = os.environ.get('LSX_AB_SOLVER', 'linear')       # 'parabolic': time and check the N4 rule instead
eng = Engine(prob, ncol, lib=lib)
eng.set_formal_solver(solver)
load(eng, blk, prof, ncol)
time_only = os.environ.get('LSX_AB_TIME_ONLY') is not None    # ablation builds (wrong results by construction): formal solution only
if not time_only:
    for _ in range(3):
        drivers.mali_step(eng)
best = (1e9, 1e9)
for _ in range(3):
    t, s = eng.time_formal_sol(2, 15)
    best = min(best, (s, t))
if time_only:
    print('%-22s %s %s ncol=%d sweep %.4f ms  fs %.4f ms  (formal solution only, results not checked)' % (os.path.basename(so), solver, wl, ncol, best[0], best[1]))
    eng.close()
    sys.exit(0)
N = 30
t0 = time.perf_counter()
for _ in range(N):
    drivers.mali_step(eng)
step = (time.perf_counter() - t0) / N * 1e3
eng.close()
# parity on 40 columns (per-class launch path) against the oracle
import oracle
ora = oracle.load()
nchk = 40
e1, e2 = Engine(prob, nchk, lib=lib, policy_columns=ncol), Engine(prob, nchk, lib=ora)      # (the mapping the timed context ran: decided for its column count)
for e in (e1, e2):
    e.set_formal_solver(solver)
    load(e, blk, prof, nchk)
ora.dll.lsx_oracle_set_threads(e2._h, 16)
e1.formal_sol_gamma(); e2.formal_sol_gamma()
eJ, eI = relerr(e1.get(_capi.LSX_J), e2.get(_capi.LSX_J)), relerr(e1.get(_capi.LSX_I), e2.get(_capi.LSX_I))
off, diag = gamma_err(e1.get(_capi.LSX_GAMMA), e2.get(_capi.LSX_GAMMA), prob)
for it in range(2, 8):
    for e in (e1, e2):
        e.formal_sol_gamma()
        if it > 3:
            e.stat_equil()
en = relerr(e1.get(_capi.LSX_N), e2.get(_capi.LSX_N))
print('%-22s %s' % (os.path.basename(so), solver), end=' ')
print('%s ncol=%d sweep %.4f ms  fs %.4f ms  step %.4f ms | call1 J %.1e I %.1e Goff %.1e Gdiag %.1e | it7 n %.1e'
      % (wl, ncol, best[0], best[1], step, eJ, eI, off, diag, en))
