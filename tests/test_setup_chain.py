"""Next row N1: the set-up chain evaluated by the library -- v_broad, damping (radiative + Unsold van der Waals
+ Stark), LTE populations, collisional rates (with scipy's not-a-knot cubic in temperature) -- behind
lsx_set_atomic_data / lsx_set_atmosphere.

Pinned on what the reference itself computed for FALC and for a perturbed atmosphere (tests/golden/setup_falc.npz, written
by make_golden.py setup from the unmodified reference: atomic_model.py:66-69, 300-345, 491-502; atomic_set.py:105-145;
collisional_rates.py:10-96).  Tolerance 1e-13 relative (rates: relative to the largest rate of the depth, the small ones
are differences of interpolated values)."""
import numpy as np
import pytest

from conftest import golden, relerr
from lightspinner_amd import fixtures, atomdata, _capi
from lightspinner_amd.problem import Engine, ColumnBlock


def _engine(lib, d, base_file='falc_cah.npz'):
    prob, block, raw = fixtures.load_problem_npz(golden(base_file))
    assert [str(x) for x in raw['atom_names']] == [str(x) for x in d['atom_names']]
    e = Engine(prob, 2, lib=lib)
    e.set_columns(0, ColumnBlock.concatenate([block, block]))
    e.set_atomic_data(atomdata.from_fixture(d))
    return e, prob, block, raw


def _atm(d, tags):
    st = lambda key: np.stack([d['%s_%s' % (t, key)] for t in tags])
    nTot = np.stack([np.stack([d['%s_a%d_nTotal' % (t, a)] for a in range(2)]) for t in tags])
    return dict(temperature=st('temperature'), ne=st('ne'), vturb=st('vturb'), nHGround=st('hGround'), nTotal=nTot)


def _check(lib, tol=1e-13):
    d = dict(np.load(golden('setup_falc.npz')))
    e, prob, block, raw = _engine(lib, d)
    tags = ('atm0', 'atm1')                         # FALC and the perturbed atmosphere, one column each
    e.set_atmosphere(0, lte_pops=True, **_atm(d, tags))
    vB, aD = e.get(_capi.LSX_VBROAD), e.get(_capi.LSX_ADAMP)
    nStar, Cr, n = e.get(_capi.LSX_NSTAR), e.get(_capi.LSX_C), e.get(_capi.LSX_N)
    for c, t in enumerate(tags):
        lo = 0
        for a in range(2):
            pre = '%s_a%d_' % (t, a)
            assert relerr(vB[c, a], d[pre + 'vBroad']) < tol
            nl = d[pre + 'aDamp'].shape[0]
            assert relerr(aD[c, lo:lo + nl], d[pre + 'aDamp']) < tol, (t, a)
            lo += nl
            o, o2 = prob.lev_off[a], prob.lev2_off[a]
            assert relerr(nStar[c, o:o + 6], d[pre + 'nStar']) < tol, (t, a)
            assert np.array_equal(n[c, o:o + 6], nStar[c, o:o + 6])             # n starts as a copy of nStar, rh_method.py:414-416
            Cref = d[pre + 'C'].reshape(36, -1)
            scale = np.abs(Cref).max(axis=0, keepdims=True)
            assert np.max(np.abs(Cr[c, o2:o2 + 36] - Cref) / scale) < tol, (t, a)
            assert np.all(Cr[c, o2:o2 + 36] >= 0.0)
    # FALC column: the profiles built from the library's own damping are the reference's (falc_cah.npz holds t.phi, t.wphi)
    assert relerr(e.get(_capi.LSX_PHI, 0, 1)[0], block.phi[0]) < 3e-13
    assert relerr(e.get(_capi.LSX_WPHI, 0, 1)[0], block.wphi[0]) < 1e-13
    # ... and the hot path runs on them like on the handed-over inputs
    dJ = e.formal_sol_gamma()
    assert dJ == 1.0
    assert relerr(e.get(_capi.LSX_I, 0, 1)[0], raw['fs1_I']) < 3e-11
    return e


def test_oracle_setup_chain_matches_the_reference(oracle_lib):
    _check(oracle_lib)


def test_setup_chain_errors(oracle_lib):
    d = dict(np.load(golden('setup_falc.npz')))
    prob, block, raw = fixtures.load_problem_npz(golden('falc_cah.npz'))
    e = Engine(prob, 1, lib=oracle_lib)
    e.set_columns(0, block)
    with pytest.raises(_capi.LsxError, match='lsx_set_atomic_data'):
        e.set_atmosphere(0, **_atm(d, ('atm0',)))
    with pytest.raises(_capi.LsxError):
        e.set_atomic_data(atomdata.from_fixture(d, atoms=[0]))              # one atom for a two-atom context


@pytest.mark.gpu
def test_hip_setup_chain_matches_the_reference_and_the_oracle(hip_lib, oracle_lib):
    eh = _check(hip_lib)
    eo = _check(oracle_lib)
    for what in (_capi.LSX_VBROAD, _capi.LSX_ADAMP, _capi.LSX_NSTAR):
        assert relerr(eh.get(what), eo.get(what)) < 1e-13
    assert relerr(eh.get(_capi.LSX_J), eo.get(_capi.LSX_J)) < 3e-11


@pytest.mark.gpu
def test_setup_chain_at_batch_size_feeds_the_hot_path(hip_lib, oracle_lib):
    """48 Ca+H columns whose broadening, damping, LTE populations, collisional rates and ray-dependent profiles all come
    from lsx_set_atmosphere on each column's own (perturbed) atmosphere -- physically consistent columns, SURVEY 8f N1 --
    then MALI iterations through the per-class launch path: HIP against the oracle"""
    from lightspinner_amd import synth
    d = dict(np.load(golden('setup_falc.npz')))
    prob, base, raw = fixtures.load_problem_npz(golden('falc_cah.npz'), phi_compact=False)
    ncol = 48
    blk, _ = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=11, vlos_sigma=0.0)       # geometry, background, placeholders
    blk.phi = blk.wphi = None
    atm = synth.perturbed_atmospheres(prob, raw, ncol, seed=21)
    assert atm['vlos'] is not None and np.any(atm['temperature'][1] != atm['temperature'][0])
    engs = []
    for lib in (hip_lib, oracle_lib):
        e = Engine(prob, ncol, lib=lib)
        e.set_columns(0, blk)
        e.set_atomic_data(atomdata.from_fixture(d))
        e.set_atmosphere(0, lte_pops=True, **atm)
        engs.append(e)
    hip, ora = engs
    oracle_lib.dll.lsx_oracle_set_threads(ora._h, 8)
    for what, tol in ((_capi.LSX_VBROAD, 1e-14), (_capi.LSX_ADAMP, 1e-13), (_capi.LSX_NSTAR, 1e-13), (_capi.LSX_WPHI, 1e-13), (_capi.LSX_PHI, 1e-12)):
        assert relerr(hip.get(what), ora.get(what)) < tol, what
    Ch, Co = hip.get(_capi.LSX_C), ora.get(_capi.LSX_C)
    assert np.max(np.abs(Ch - Co) / np.abs(Co).max(axis=1, keepdims=True)) < 1e-13
    # column 0 is the unperturbed FALC atmosphere: the reference's own numbers
    assert relerr(hip.get(_capi.LSX_NSTAR, 0, 1)[0], np.concatenate([d['atm0_a%d_nStar' % a] for a in range(2)])) < 1e-13
    for it in range(1, 7):
        assert hip.formal_sol_gamma() == pytest.approx(ora.formal_sol_gamma(), rel=1e-7)
        if it == 1:
            assert relerr(hip.get(_capi.LSX_J), ora.get(_capi.LSX_J)) < 3e-11 and relerr(hip.get(_capi.LSX_I), ora.get(_capi.LSX_I)) < 3e-11
        if it > 3:
            assert hip.stat_equil() == pytest.approx(ora.stat_equil(), rel=1e-6)
    assert relerr(hip.get(_capi.LSX_N), ora.get(_capi.LSX_N)) < 1e-8
