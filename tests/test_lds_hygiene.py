"""Uninitialised LDS (round 5).  LDS keeps its contents from one kernel to the next and an idle GPU's LDS is mostly zeros, so a kernel
that reads a cell it never wrote usually looks correct -- until it follows a kernel that left something else there.  Here every CU's LDS
is filled with NaN (lsx_hip_poison_lds, a diagnostic entry of the HIP library) before each formal solution, on every launch path and
mapping: any read of an unwritten cell that reaches a result turns it into NaN, and the comparison with the oracle fails.
(What it caught when it was written: a tile WITHOUT fast continua inside a class that runs the folded ray-serial instance read four
rows of cross-sections nobody had written, multiplied them with zeros -- NaN x 0 -- and showed up as a singular statistical
equilibrium once in a dozen runs.)"""
import ctypes as C

import numpy as np
import pytest

from conftest import golden, relerr, gamma_err
from lightspinner_amd import fixtures, synth, Engine, _capi

pytestmark = pytest.mark.gpu


def _poison(hip_lib):
    f = hip_lib.dll.lsx_hip_poison_lds
    f.argtypes = [C.c_int32, C.c_int32]
    assert f(0, 3) == 0


@pytest.mark.parametrize('solver', ['linear', 'parabolic'])
@pytest.mark.parametrize('name,ncol,policy,tol', [('falc_cah.npz', 41, 'ray-serial', 3e-11), ('falc_cah.npz', 41, 'ray-per-lane', 3e-11),
                                                  ('falc_ca.npz', 36, 'ray-serial', 1e-12), ('falc_cah.npz', 3, 'auto', 3e-11),
                                                  ('falc_mg.npz', 33, 'ray-serial', 3e-11)])
def test_results_do_not_depend_on_what_the_previous_kernel_left_in_lds(hip_lib, oracle_lib, name, ncol, policy, tol, solver):
    prob, base, raw = fixtures.load_problem_npz(golden(name), phi_compact=False)
    blk, prof = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=31, vlos_sigma=2.0e3)
    hip = Engine(prob, ncol, lib=hip_lib, sweep_policy=policy)
    ora = Engine(prob, ncol, lib=oracle_lib)
    for e in (hip, ora):
        synth.load_columns(e, blk, prof)
        e.set_formal_solver(solver)
    oracle_lib.dll.lsx_oracle_set_threads(ora._h, 8)
    for it in range(1, 6):
        _poison(hip_lib)
        dJ, dJo = hip.formal_sol_gamma(), ora.formal_sol_gamma()
        assert np.isfinite(dJ) and dJ == pytest.approx(dJo, rel=1e-6)
        for w in (_capi.LSX_J, _capi.LSX_I, _capi.LSX_GAMMA):
            assert np.isfinite(hip.get(w)).all(), (it, w)
        if it == 1:
            assert relerr(hip.get(_capi.LSX_J), ora.get(_capi.LSX_J), floor=1e-300) < tol and relerr(hip.get(_capi.LSX_I), ora.get(_capi.LSX_I)) < tol
            off, diag = gamma_err(hip.get(_capi.LSX_GAMMA), ora.get(_capi.LSX_GAMMA), prob)
            assert off < 10 * tol and diag < tol, (off, diag)
        if it > 3:
            _poison(hip_lib)
            assert hip.stat_equil() == pytest.approx(ora.stat_equil(), rel=1e-6)
    assert relerr(hip.get(_capi.LSX_N), ora.get(_capi.LSX_N)) < 1e-7
    hip.close(); ora.close()
