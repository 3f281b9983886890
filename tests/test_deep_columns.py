"""Depth grids beyond the reference's 82 points on the ray-serial mapping (round 5).  Until round 4 the ray-serial kernel staged the
per-depth operands of its five columns for the WHOLE column in LDS, so contexts of more than ~160 depths fell back to one ray per
lane (lsx_plan.cpp: rs_ok); now the operands come through a ring of eight rows per wave fed from a table in HBM (lsx_plan.h, "a RING in
LDS"; lsx_hip.hip, k_build_optab) and the LDS of a workgroup does not depend on Nspace.  formal_solver.py:95-139 takes any Nspace.
Columns: FALC-perturbed CaII / Ca+H columns on the 4x and 8x refined grids of tests/parabolic_cases.py (325 and 649 depths; every
input interpolated monotonically over the depth index), seven columns = one full wavefront group and one of two; HIP on the pinned
ray-serial mapping against the oracle at the tolerances of the 82-depth tests."""
import numpy as np
import pytest

from conftest import golden, relerr, gamma_err
from lightspinner_amd import fixtures, synth, Engine, _capi
from parabolic_cases import _refine_depth
from test_production_classes import class_table

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,factor,tol,solver', [('falc_ca.npz', 4, 1e-12, 'linear'), ('falc_cah.npz', 4, 3e-11, 'linear'),
                                                    ('falc_ca.npz', 8, 1e-12, 'linear'), ('falc_cah.npz', 4, 3e-11, 'parabolic')])
def test_deep_columns_run_the_ray_serial_kernel_and_meet_the_oracle(hip_lib, oracle_lib, name, factor, tol, solver):
    prob, base, raw = fixtures.load_problem_npz(golden(name))
    batch, _ = synth.perturbed_columns(prob, base, raw, ncol=7, seed=4242, vlos_sigma=0.0)
    fine, fblock, xf = _refine_depth(prob, batch, factor)
    assert fine.Nspace == factor * 81 + 1 and fblock.ncol == 7
    hip = Engine(fine, 7, lib=hip_lib, sweep_policy='ray-serial')
    ora = Engine(fine, 7, lib=oracle_lib)
    for e in (hip, ora):
        e.set_columns(0, fblock)
        e.set_formal_solver(solver)
    oracle_lib.dll.lsx_oracle_set_threads(ora._h, 7)
    assert hip.sweep_policy() == 'ray-serial'
    kv = dict(x.split('=', 1) for x in hip.effective_options().split(';'))
    assert kv['mapping'] == 'ray-serial' and 's' in kv['classes']
    for it in range(1, 6):
        dJ, dJo = hip.formal_sol_gamma(), ora.formal_sol_gamma()
        tight = it < 5
        assert dJ == pytest.approx(dJo, rel=1e-9 if tight else 1e-6)
        assert relerr(hip.get(_capi.LSX_J), ora.get(_capi.LSX_J)) < (tol if tight else 1e-9)
        assert relerr(hip.get(_capi.LSX_I), ora.get(_capi.LSX_I)) < (tol if tight else 1e-9)
        off, diag = gamma_err(hip.get(_capi.LSX_GAMMA), ora.get(_capi.LSX_GAMMA), fine)
        assert off < (10 * tol if tight else 1e-7) and diag < (tol if tight else 1e-8), (it, off, diag)
        if it > 3:
            assert hip.stat_equil() == pytest.approx(ora.stat_equil(), rel=1e-6)
    # what ran: per-class launches, every class with at most two per-ray slots on the ray-serial kernel (the parabolic rule: its
    # classes of at most one line)
    table, fused = class_table(hip_lib, hip)
    assert fused == 0 and all(launches == 5 for _, launches in table.values())
    serial = {k for k, v in class_table.ray_serial.items() if v}
    if solver == 'linear':
        assert serial == {k for k in table if 0 <= k[0] <= 2}
    else:
        assert serial == {k for k in table if k in ((0, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 0), (2, 2, 0, 1), (2, 2, 0, 2))} and serial
    hip.close(); ora.close()
