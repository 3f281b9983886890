"""The EPI instances of the ray-serial sweep (round 5; lsx_plan.h, "their GAMMA INTEGRANDS"): with `epi=1` the wave that visits a depth
second forms the Gamma integrands of the tile's fast continua and the linked lines' corrections itself (rh_method.py:652, 677-681 for
ray-independent transitions -- what k_fast_gamma_cols does otherwise) and no fast-continuum epilogue is launched for those classes.
The option is OFF by default: measured 2 % slower than the kernel it replaces (profiles/r05/ab_epilogue_in_sweep.txt).  It stays a
tested part of the library: FALC Ca+H and CaII production columns, the reference's MgII atom, a toy problem with an odd depth count
(the two directions meet in ONE step: the midpoint's exchange), frozen columns, every CU's LDS poisoned before each call -- all
against the oracle at the tolerances of the default path, and against the default path itself (same terms, other association)."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden, relerr, gamma_err
from lightspinner_amd import fixtures, synth, Engine, _capi
from toy import toy_problem

pytestmark = pytest.mark.gpu


def _poison(hip_lib):
    f = hip_lib.dll.lsx_hip_poison_lds
    f.argtypes = [C.c_int32, C.c_int32]
    assert f(0, 2) == 0


def _classes(e):
    return dict(x.split('=', 1) for x in e.effective_options().split(';'))['classes']


@pytest.mark.parametrize('name,ncol,tol,ntol', [('falc_cah.npz', 41, 3e-11, 1e-8), ('falc_ca.npz', 36, 1e-12, 1e-8), ('falc_mg.npz', 33, 3e-11, 1e-7)])
def test_epilogue_in_the_sweep_meets_the_oracle(hip_lib, oracle_lib, name, ncol, tol, ntol):
    prob, base, raw = fixtures.load_problem_npz(golden(name), phi_compact=False)
    blk, prof = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=2025, vlos_sigma=2.0e3)
    epi = Engine(prob, ncol, lib=hip_lib, sweep_policy='ray-serial', options='epi=1')
    std = Engine(prob, ncol, lib=hip_lib, sweep_policy='ray-serial')
    ora = Engine(prob, ncol, lib=oracle_lib)
    for e in (epi, std, ora):
        synth.load_columns(e, blk, prof)
    oracle_lib.dll.lsx_oracle_set_threads(ora._h, 8)
    assert 'sfe' in _classes(epi) and 'sfe' not in _classes(std) and 'sf' in _classes(std)
    assert epi.options_signature() != std.options_signature()
    for it in range(1, 8):
        _poison(hip_lib)
        dJ, dJs, dJo = epi.formal_sol_gamma(), std.formal_sol_gamma(), ora.formal_sol_gamma()
        assert dJ == pytest.approx(dJo, rel=1e-6) and dJ == pytest.approx(dJs, rel=1e-9)
        if it == 1:
            assert relerr(epi.get(_capi.LSX_J), ora.get(_capi.LSX_J), floor=1e-300) < tol and relerr(epi.get(_capi.LSX_I), ora.get(_capi.LSX_I)) < tol
            off, diag = gamma_err(epi.get(_capi.LSX_GAMMA), ora.get(_capi.LSX_GAMMA), prob)
            assert off < 10 * tol and diag < tol, (off, diag)
            # J and I do not involve the epilogue at all: the same bits as the default path; Gamma: the same terms summed in another order
            assert np.array_equal(epi.get(_capi.LSX_J), std.get(_capi.LSX_J)) and np.array_equal(epi.get(_capi.LSX_I), std.get(_capi.LSX_I))
            off, diag = gamma_err(epi.get(_capi.LSX_GAMMA), std.get(_capi.LSX_GAMMA), prob)
            assert 0 < off < 1e-11 and diag < 1e-12, (off, diag)
        if it > 3:
            dP, dPo = epi.stat_equil(), ora.stat_equil()
            std.stat_equil()
            assert dP == pytest.approx(dPo, rel=1e-6)
    assert relerr(epi.get(_capi.LSX_N), ora.get(_capi.LSX_N)) < ntol
    # every second column frozen: its slabs, Gamma and populations stay (the in-sweep epilogue skips frozen columns like the kernel)
    mask = np.arange(ncol) % 2 == 0
    before = {w: epi.get(w) for w in (_capi.LSX_GAMMA, _capi.LSX_N, _capi.LSX_J)}
    for e in (epi, ora):
        e.set_active_columns(mask)
        e.formal_sol_gamma(); e.stat_equil()
    for w, v in before.items():
        assert np.array_equal(epi.get(w)[~mask], v[~mask])
    assert relerr(epi.get(_capi.LSX_N)[mask], ora.get(_capi.LSX_N)[mask]) < ntol
    for e in (epi, std, ora):
        e.close()


@pytest.mark.parametrize('kw', [dict(seed=13, Nrays=5, Nspace=41, Nspect=120, ncol=34, multiplet=3), dict(seed=11, Nrays=5, Nspace=82, Nspect=140, ncol=33, chain=False),
                                dict(seed=21, Nrays=5, Nspace=3, Nspect=60, ncol=36)], ids=['odd-depths', 'even-depths', 'three-depths'])
def test_epilogue_in_the_sweep_on_toy_topologies(hip_lib, oracle_lib, kw):
    kw = dict(kw)
    ncol = kw['ncol']
    prob, block = toy_problem(**kw)
    eh, eo = Engine(prob, ncol, lib=hip_lib, sweep_policy='ray-serial', options='epi=1'), Engine(prob, ncol, lib=oracle_lib)
    for e in (eh, eo):
        e.set_columns(0, block)
    for it in range(6):
        _poison(hip_lib)
        dh, do = eh.formal_sol_gamma(), eo.formal_sol_gamma()
        if it == 0:
            off, diag = gamma_err(eh.get(_capi.LSX_GAMMA), eo.get(_capi.LSX_GAMMA), prob)
            assert off < 1e-10 and diag < 1e-11, (off, diag)
        assert abs(dh - do) <= 1e-7 * max(abs(do), 1e-3)
        if it >= 2:
            eh.stat_equil(); eo.stat_equil()
    n_o = eo.get(_capi.LSX_N)
    dn = np.abs(eh.get(_capi.LSX_N) - n_o) / np.abs(n_o).max(axis=1, keepdims=True)
    assert dn.max() < 1e-8
    eh.close(); eo.close()
