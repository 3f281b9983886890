"""Next row N1: ComputationalTransition.compute_phi (rh_method.py:198-243) evaluated by the library.

Pinned on the profiles the reference itself produced (the golden fixtures hold t.phi / t.wphi next to the aDamp /
vBroad / vlos they were computed from), and on scipy.special.wofz, which is what the reference calls (utils.py:13-15).
Tolerance 1e-13 relative on phi and wphi (measured: 3e-14 / 3e-15)."""
import ctypes as C

import numpy as np
import pytest
from scipy.special import wofz

from conftest import golden, relerr
import refprofile
from lightspinner_amd import fixtures, _capi
from lightspinner_amd.problem import Engine

CASES = [('falc_ca.npz', True), ('falc_ca_vlos.npz', False), ('falc_cah.npz', True), ('falc_ca.npz', False)]


def _inputs(prob, raw, compact):
    return fixtures.profile_inputs(prob, raw, with_vlos=not compact)


def _check(lib, name, compact):
    prob, block, raw = fixtures.load_problem_npz(golden(name), phi_compact=compact)
    aD, vB, vlos = _inputs(prob, raw, compact)
    e = Engine(prob, 1, lib=lib)
    bare = block.slice(0, 1)
    bare.phi = bare.wphi = None                     # profiles are NOT handed over
    e.set_columns(0, bare)
    e.set_line_profiles(0, aD, vB, vlos)
    phi, wphi = e.get(_capi.LSX_PHI)[0], e.get(_capi.LSX_WPHI)[0]
    if block.phi is not None:
        assert relerr(phi, block.phi[0]) < 1e-13
        assert relerr(wphi, block.wphi[0]) < 1e-13
    else:                                           # vlos != 0: the file holds a strided sample of the reference's phi
        o = 0
        for li, (kr, t) in enumerate([(kr, t) for kr, t in enumerate(prob.trans) if t.is_line]):
            assert relerr(phi[o:o + t.Nlambda][::7, :, :, ::9], raw['t%d_phi_sample' % kr]) < 1e-13
            assert relerr(wphi[li], raw['t%d_wphi' % kr]) < 1e-13
            o += t.Nlambda
    return e, prob, block, raw


def test_oracle_voigt_against_scipy(oracle_lib):
    f = oracle_lib.dll.lsx_oracle_voigt
    f.restype = C.c_double
    f.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(7)
    for a in (1e-4, 1e-3, 1e-2, 0.1, 1.0, 3.0):
        v = np.concatenate([np.linspace(-12, 12, 2001), np.logspace(1, 3.5, 200), rng.uniform(0, 8, 500)])
        ref = wofz(v + 1j * a).real
        got = np.array([f(a, x) for x in v])
        assert np.max(np.abs(got - ref) / ref) < 1e-13, a
    # a -> 0: the Gaussian core and the 1/v^2 wing, to the accuracy scipy itself has there
    v = np.array([0.0, 0.3, 1.7, 4.0, 30.0])
    assert np.max(np.abs(np.array([f(1e-7, x) for x in v]) / wofz(v + 1e-7j).real - 1.0)) < 1e-9


@pytest.mark.parametrize('name,compact', CASES)
def test_oracle_profiles_match_the_reference(oracle_lib, name, compact):
    _check(oracle_lib, name, compact)


def test_perturbed_profile_inputs_against_scipy(oracle_lib):
    # tests/refprofile.py (scipy wofz, the routine the reference calls) on a column with vlos != 0 and scaled damping
    prob, block, raw = fixtures.load_problem_npz(golden('falc_ca.npz'), phi_compact=False)
    rng = np.random.default_rng(3)
    Ns = prob.Nspace
    vlos = 3e3 * np.sin(np.linspace(0, 5, Ns)) + rng.normal(0, 300, Ns)
    aD, vB, _ = _inputs(prob, raw, False)
    aD = aD * rng.uniform(0.5, 2.0, aD.shape)
    e = Engine(prob, 1, lib=oracle_lib)
    e.set_columns(0, block)
    e.set_line_profiles(0, aD, vB, vlos[None])
    phi, wphi = e.get(_capi.LSX_PHI)[0], e.get(_capi.LSX_WPHI)[0]
    o = 0
    for li, (kr, t) in enumerate([(kr, t) for kr, t in enumerate(prob.trans) if t.is_line]):
        ph, wp = refprofile.profiles(raw['t%d_wavelength' % kr], t.lambda0, aD[0, li], vB[0, t.atom], vlos, prob.muz, prob.wmu)
        assert relerr(phi[o:o + t.Nlambda], ph) < 1e-13
        assert relerr(wphi[li], wp) < 1e-13
        o += t.Nlambda


@pytest.mark.gpu
@pytest.mark.parametrize('name,compact', CASES)
def test_hip_profiles_match_the_reference(hip_lib, oracle_lib, name, compact):
    e, prob, block, raw = _check(hip_lib, name, compact)
    # and the hot path runs on them: one call against the oracle fed with the reference's own profiles
    o = Engine(prob, 1, lib=oracle_lib)
    o.set_columns(0, block)
    if block.phi is None:
        o.set_line_profiles(0, *_inputs(prob, raw, compact))
    assert e.formal_sol_gamma() == pytest.approx(o.formal_sol_gamma(), rel=1e-9)
    assert relerr(e.get(_capi.LSX_J), o.get(_capi.LSX_J)) < 1e-11
    assert relerr(e.get(_capi.LSX_I), o.get(_capi.LSX_I)) < 1e-11


@pytest.mark.gpu
def test_hip_profiles_many_columns(hip_lib, oracle_lib):
    prob, block, raw = fixtures.load_problem_npz(golden('falc_ca.npz'), phi_compact=False)
    rng = np.random.default_rng(5)
    ncol, Ns = 37, prob.Nspace
    aD, vB, _ = _inputs(prob, raw, False)
    aD = aD * rng.uniform(0.3, 3.0, (ncol,) + aD.shape[1:])
    vB = vB * rng.uniform(0.8, 1.25, (ncol,) + vB.shape[1:])
    vlos = rng.normal(0, 2e3, (ncol, Ns))
    outs = []
    for lib in (hip_lib, oracle_lib):
        e = Engine(prob, ncol, lib=lib)
        from lightspinner_amd.problem import ColumnBlock
        bare = ColumnBlock.concatenate([block] * ncol)
        bare.phi = bare.wphi = None
        e.set_columns(0, bare)
        e.set_line_profiles(0, aD, vB, vlos)
        outs.append((e.get(_capi.LSX_PHI), e.get(_capi.LSX_WPHI)))
    assert relerr(outs[0][0], outs[1][0]) < 1e-13
    assert relerr(outs[0][1], outs[1][1]) < 1e-13
