"""Drop-in check against the LIVE reference (build container only: /root/reference is absent on the
GPU box, where this file skips).  The reference's own objects are handed to both Contexts."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, relerr

REF = os.environ.get('LIGHTSPINNER_REF', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference not mounted')


def test_context_against_live_reference(oracle_lib):
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    argv, sys.argv = sys.argv, ['x']
    try:
        import make_golden as mg          # sets up the stand-ins and imports the reference modules
    finally:
        sys.argv = argv
    from lightspinner_amd.rh_method import Context

    ref = mg.build_ctx(['Ca'])
    # a second, independent set of reference objects for our Context (Context mutates eqPops in place)
    mine_src = mg.build_ctx(['Ca'])
    mine = Context(mine_src.atmos, mine_src.spect, mine_src.eqPops, mine_src.background, lib=oracle_lib)
    assert mine.setup == 'native'         # the reference's model objects carry their atomic data: the library's set-up chain
    assert mine.activeAtoms[0].n is mine_src.eqPops['Ca'].n
    assert relerr(mine.activeAtoms[0].vBroad, ref.activeAtoms[0].vBroad) < 1e-14
    for t_ref, t_mine in zip(ref.activeAtoms[0].trans, mine.activeAtoms[0].trans):
        assert t_ref.Nblue == t_mine.Nblue and np.array_equal(t_ref.active, t_mine.active)
        if t_ref.isLine:
            aD = t_ref.transModel.damping(ref.atmos, ref.activeAtoms[0].vBroad, ref.activeAtoms[0].hPops.n[0])[0]
            assert relerr(t_mine.aDamp, aD) < 1e-13 and np.array_equal(t_mine.wlambda(), t_ref.wlambda())
            assert relerr(t_mine.phi, t_ref.phi) < 1e-13 and relerr(t_mine.wphi, t_ref.wphi) < 1e-13
    for it in range(1, 7):
        dJ_ref, dJ = ref.formal_sol_gamma_matrices(), mine.formal_sol_gamma_matrices()
        assert dJ == pytest.approx(dJ_ref, rel=1e-8)
        tol = 1e-12 if it < 5 else 1e-8
        assert relerr(mine.J, ref.J) < tol and relerr(mine.I, ref.I) < tol
        assert relerr(mine.activeAtoms[0].C, ref.activeAtoms[0].C, floor=1e-300) < 1e-13
        if it > 3:
            dP_ref, dP = ref.stat_equil(), mine.stat_equil()
            assert dP == pytest.approx(dP_ref, rel=1e-7)
            assert relerr(mine_src.eqPops['Ca'].n, ref.eqPops['Ca'].n) < 1e-7


def test_wavelength_grid_against_live_reference(oracle_lib):
    """the reference's RadiativeSet.compute_wavelength_grid and lightspinner_amd.spectrum.compute_wavelength_grid on two
    independent sets of the reference's own model objects (CaII + H active): grid, blueIdx and active sets identical,
    continuum alpha to 1e-14"""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    argv, sys.argv = sys.argv, ['x']
    try:
        import make_golden as mg
    finally:
        sys.argv = argv
    from lightspinner_amd.spectrum import compute_wavelength_grid
    aSet = mg.RadiativeSet([mg.CaII_atom(), mg.H_6_atom()])
    aSet.set_active('Ca', 'H')
    ref = aSet.compute_wavelength_grid()
    fresh = {m.name: m for m in mg.RadiativeSet([mg.CaII_atom(), mg.H_6_atom()]).atoms}
    mine = compute_wavelength_grid([fresh[m.name] for m in ref.models], lib=oracle_lib)      # same model order as the reference's set
    assert np.array_equal(mine.wavelength, ref.wavelength)
    assert mine.blueIdx == [int(b) for b in ref.blueIdx]
    assert len(mine.transitions) == len(ref.transitions)
    for t_ref, t_mine in zip(ref.transitions, mine.transitions):
        assert (t_ref.i, t_ref.j, t_ref.atom.name) == (t_mine.i, t_mine.j, t_mine.atom.name)
        assert np.array_equal(t_ref.wavelength, t_mine.wavelength)
        if not isinstance(t_ref, mg.AtomicLine):
            assert np.max(np.abs(t_ref.alpha - t_mine.alpha)) <= 1e-14 * np.max(t_ref.alpha)
    for la in range(ref.wavelength.shape[0]):
        assert [(t.atom.name, t.i, t.j) for t in ref.activeSet[la]] == [(t.atom.name, t.i, t.j) for t in mine.activeSet[la]]
