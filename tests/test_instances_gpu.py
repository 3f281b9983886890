"""GPU side of the instance ledger (tests/instance_cases.py, tests/test_instance_ledger.py): every topology case on
  lane      the one-ray-per-lane kernels (lsx_sweep.hip), one launch per tile class
  serial    the ray-serial kernels (lsx_sweep_rs.hip) for the classes that have an instance there, one ray per lane for the rest
  parabolic the parabolic rule's compile-time tile classes (sweep_tile_par) and its generic instance
  parabolic-serial  the parabolic rule on the ray-serial kernels (lsx_sweep_rs.hip, PAR) for the classes that have an instance there
against the oracle on the same columns (rh_method.py:595-692, formal_solver.py:14-212; the topologies: rh_method.py:606-627,
654-681 on atoms shaped like rh_atoms.py:194, :355), and the classes that ran are the ones the case is in the ledger for.

Tolerances as tests/test_toy_topologies.py: one formal solution 1e-11 on I, J (2e-10 where the case has an interval just above the
5e-4 switch of w2), off-diagonal Gamma 1e-10, diagonal 1e-11 of its column's largest entry; populations after six
iterations 1e-8 of the depth's largest population."""
import ctypes as C

import numpy as np
import pytest

import envelope
from conftest import relerr, gamma_err
import instance_cases as ic
from lightspinner_amd import _capi
from lightspinner_amd.problem import Engine

pytestmark = pytest.mark.gpu


def classes_run(lib, eng):
    """-> {(slots, lines, linked, relation): (launches, on the ray-serial kernel)}, fused launches"""
    f = lib.dll.lsx_hip_class_info
    f.restype = C.c_int32
    f.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    out = (C.c_int64 * 8)()
    n = f(eng._h, -1, out)
    fused = int(out[0])
    table = {}
    for i in range(n):
        f(eng._h, i, out)
        table[(int(out[0]), int(out[1]), int(out[4]), int(out[5]))] = (int(out[3]), bool(out[6]))
    return table, fused


@pytest.mark.parametrize('mode', ['lane', 'serial', 'parabolic', 'parabolic-serial'])
@pytest.mark.parametrize('case', ic.CASES, ids=[c[0] for c in ic.CASES])
def test_case_meets_the_oracle(hip_lib, oracle_lib, case, mode):
    name, ncol, Ns, compact = case
    prob, block = ic.build(name, ncol, Ns, compact)
    eh = Engine(prob, ncol, lib=hip_lib, sweep_policy='ray-serial' if mode.endswith('serial') else 'ray-per-lane')
    eo = Engine(prob, ncol, lib=oracle_lib)
    oracle_lib.dll.lsx_oracle_set_threads(eo._h, 8)
    for e in (eh, eo):
        e.set_columns(0, block)
        e.set_formal_solver('parabolic' if mode.startswith('parabolic') else 'linear')
    # (crowd: 2.17e-10 on the emergent intensity of one ray -- the same value on both kernels -- with J at 1.4e-13 and Gamma at 5e-15:
    # the signature of an interval just above the 5e-4 switch of w2, where w1 = (1 - e) - dtau e cancels to dtau^2 / 2.  Round 4 set
    # 5e-10 / 2e-10 from such measurements; now every entry has to lie inside 1e-11 + 3 x what a one-ulp change of the oracle's OWN
    # exp() does to that entry: tests/envelope.py)
    for it in range(6):
        dh, do = eh.formal_sol_gamma(), eo.formal_sol_gamma()
        if it == 0:
            envelope.first_call_inside(oracle_lib, prob, block, eh.get(_capi.LSX_I), eh.get(_capi.LSX_J),
                                       solver='parabolic' if mode.startswith('parabolic') else 'linear')
            off, diag = gamma_err(eh.get(_capi.LSX_GAMMA), eo.get(_capi.LSX_GAMMA), prob)
            assert off < 1e-10 and diag < 1e-11, (off, diag)
        assert abs(dh - do) <= 1e-7 * max(abs(do), 1e-3)
        if it >= 2:
            ph, po = eh.stat_equil(), eo.stat_equil()
            assert abs(ph - po) <= 1e-6 * max(abs(po), 1e-3)
    n_o = eo.get(_capi.LSX_N)
    dn = np.abs(eh.get(_capi.LSX_N) - n_o) / np.abs(n_o).max(axis=1, keepdims=True)
    assert dn.max() < 1e-8 and relerr(eh.get(_capi.LSX_J), eo.get(_capi.LSX_J)) < 1e-7
    # ---- the ledger: the expected classes ran, each on the kernel this mode is about
    table, fused = classes_run(hip_lib, eh)
    assert fused == 0
    for key in ic.EXPECT[name]:
        assert key in table and table[key][0] == 6, (key, table)
    serial = {k for k, (_, rs) in table.items() if rs}
    if mode == 'serial':
        assert serial == {k for k in table if 0 <= k[0] <= 2}, (serial, table)      # every class with at most two per-ray slots
    elif mode == 'parabolic-serial':
        assert serial == {k for k in table if k in ic.PARABOLIC_SERIAL}, (serial, table)
    else:
        assert not serial
    eh.close(); eo.close()


@pytest.mark.parametrize('mode', ['lane', 'serial', 'parabolic', 'parabolic-serial'])
@pytest.mark.parametrize('Ns', [3, 4, 5, 6])
def test_shallow_columns_meet_the_oracle(hip_lib, oracle_lib, Ns, mode):
    """the smallest atmospheres lsx_create admits (Nspace >= 3, formal_solver.py:120-139): the first point, the meeting point of the two
    directions and the end point fall into neighbouring steps -- three depths: the midpoint's step is the last one; four: no midpoint --
    on every mapping and both rules (the parabolic rule runs one depth behind the opacities: its last request must not leave the column)"""
    prob, block = ic.build('two_atoms', 33, Ns, False)
    eh = Engine(prob, 33, lib=hip_lib, sweep_policy='ray-serial' if mode.endswith('serial') else 'ray-per-lane')
    eo = Engine(prob, 33, lib=oracle_lib)
    for e in (eh, eo):
        e.set_columns(0, block)
        e.set_formal_solver('parabolic' if mode.startswith('parabolic') else 'linear')
    solver = 'parabolic' if mode.startswith('parabolic') else 'linear'

    def make():
        e = Engine(prob, 33, lib=oracle_lib)
        e.set_columns(0, block)
        e.set_formal_solver(solver)
        return e
    runs = envelope.oracle_runs(oracle_lib, make, 4, se_from=1, what=(_capi.LSX_I, _capi.LSX_J))     # the oracle as it is and with exp() a ulp up / down
    for it in range(4):
        dh, do = eh.formal_sol_gamma(), eo.formal_sol_gamma()
        # inside the one-ulp-exp envelope entry by entry (tests/envelope.py; round 4: 2e-10 throughout, from a measurement); after the
        # first statistical equilibrium the two sides' populations differ by the LU's rounding x conditioning: 2e-10 as the floor then
        for w in (_capi.LSX_I, _capi.LSX_J):
            envelope.inside(eh.get(w), runs, it, w, 1e-11 if it < 2 else 2e-10)
        off, diag = gamma_err(eh.get(_capi.LSX_GAMMA), eo.get(_capi.LSX_GAMMA), prob)
        assert off < 1e-10 and diag < 1e-11, (it, off, diag)
        assert abs(dh - do) <= 1e-7 * max(abs(do), 1e-3)
        if it >= 1:
            ph, po = eh.stat_equil(), eo.stat_equil()
            assert abs(ph - po) <= 1e-6 * max(abs(po), 1e-3)
    table, fused = classes_run(hip_lib, eh)
    serial = {k for k, (_, rs) in table.items() if rs}
    assert fused == 0 and bool(serial) == mode.endswith('serial')
    eh.close(); eo.close()


@pytest.mark.parametrize('ncol', [3, 40])
def test_atom_without_radiative_transitions(hip_lib, oracle_lib, ncol):
    """three atoms, the middle one with collisions only: its Gamma is its C with the diagonal of rh_method.py:698-703 -- the Gamma epilogue
    runs per (column, depth, atom) and that atom's slot list is empty (fewer than 32 columns: the small-batch epilogue)"""
    from toy import spec_problem
    atoms = [(3, [('l', 0, 1, 0.1, 0.5), ('c', 1, 2, 0.0, 0.3)]), (2, []), (3, [('l', 0, 2, 0.4, 0.9), ('c', 0, 2, 0.0, 0.2)])]
    prob, block = spec_problem(atoms, seed=5, Nspace=30, Nrays=5, Nspect=200, ncol=ncol)
    eh, eo = Engine(prob, ncol, lib=hip_lib), Engine(prob, ncol, lib=oracle_lib)
    for e in (eh, eo):
        e.set_columns(0, block)
    for it in range(4):
        dh, do = eh.formal_sol_gamma(), eo.formal_sol_gamma()
        off, diag = gamma_err(eh.get(_capi.LSX_GAMMA), eo.get(_capi.LSX_GAMMA), prob)
        assert abs(dh - do) <= 1e-9 * max(abs(do), 1e-3) and off < 1e-10 and diag < 1e-11
        ph, po = eh.stat_equil(), eo.stat_equil()
        assert abs(ph - po) <= 1e-6 * max(abs(po), 1e-3)
    assert relerr(eh.get(_capi.LSX_N), eo.get(_capi.LSX_N)) < 1e-9
    eh.close(); eo.close()
