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


def test_switching_mapping_and_rule_on_one_context_gives_each_configuration_its_own_bits(hip_lib):
    """One context, 36 FALC Ca+H columns: the sweep policy and the rule are switched between formal solutions -- ray-serial (folded
    instances, operand table), one ray per lane, the parabolic rule on both mappings, back again.  Every call gives exactly the bits a
    fresh context pinned to that configuration gives from the same populations: nothing of one mapping's state (the operand table and
    its freshness, the effective-background streams of the pre-pass, the per-direction sums) leaks into the next."""
    prob, base, raw = fixtures.load_problem_npz(golden('falc_cah.npz'), phi_compact=False)
    blk, prof = synth.perturbed_columns(prob, base, raw, ncol=36, seed=808, vlos_sigma=2.0e3)
    seq = [('linear', 'ray-serial'), ('linear', 'ray-per-lane'), ('parabolic', 'ray-serial'), ('linear', 'ray-serial'),
           ('parabolic', 'ray-per-lane'), ('linear', 'ray-per-lane'), ('linear', 'ray-serial')]
    one = Engine(prob, 36, lib=hip_lib)
    synth.load_columns(one, blk, prof)
    got = []
    for i, (rule, policy) in enumerate(seq):
        one.set_formal_solver(rule)
        one.set_sweep_policy(policy)
        one.formal_sol_gamma()
        if i % 2 == 1:
            one.stat_equil()                  # the populations move: the operand table has to follow
        got.append({w: one.get(w) for w in (_capi.LSX_J, _capi.LSX_I, _capi.LSX_GAMMA, _capi.LSX_N)})
    # replay: a fresh context per step, started from the state the one context had before that step
    n_before, J_before = blk.n.copy(), None
    for i, (rule, policy) in enumerate(seq):
        e = Engine(prob, 36, lib=hip_lib, sweep_policy=policy)
        synth.load_columns(e, blk, prof)
        e.set(_capi.LSX_N, n_before)
        if J_before is not None:
            e.set(_capi.LSX_J, J_before)
        e.set_formal_solver(rule)
        e.formal_sol_gamma()
        if i % 2 == 1:
            e.stat_equil()
        for w in (_capi.LSX_J, _capi.LSX_I, _capi.LSX_GAMMA, _capi.LSX_N):
            assert np.array_equal(e.get(w), got[i][w]), (i, rule, policy, w)
        n_before, J_before = got[i][_capi.LSX_N], got[i][_capi.LSX_J]
        e.close()
    one.close()
