"""Pin the oracle (oracle/lsx_oracle.c) against golden vectors produced by importing
the unmodified reference (tests/golden/make_golden.py).  CPU only."""
import ctypes as C

import numpy as np
import pytest

import envelope

from conftest import golden, relerr, gamma_err
from lightspinner_amd import fixtures, Engine, _capi, drivers
from lightspinner_amd._capi import _ptr


class EngineAdapter:
    def __init__(self, eng):
        self.eng = eng
    def formal_sol_gamma_matrices(self):
        return self.eng.formal_sol_gamma()
    def stat_equil(self):
        return self.eng.stat_equil()


def test_w2_all_branches(oracle_lib):
    d = np.load(golden('units.npz'))
    w = np.zeros(2)
    for x, ref in zip(d['w2_dtau'], d['w2_out']):
        oracle_lib.dll.lsx_oracle_w2(float(x), _ptr(w))
        # np.exp and libm exp may differ by an ulp of exp(-dtau) <= 1; w0 = 1 - e and
        # w1 = w0 - dtau e inherit that as an ABSOLUTE error (the formulas cancel)
        assert np.allclose(w, ref, rtol=4e-16, atol=5e-16), (x, w, ref)


def test_planck(oracle_lib):
    d = np.load(golden('units.npz'))
    for wi, wav in enumerate(d['planck_wav']):
        for ti, T in enumerate(d['planck_T']):
            assert oracle_lib.dll.lsx_oracle_planck(float(T), float(wav)) == pytest.approx(d['planck_B'][wi, ti], rel=1e-14)


def test_piecewise_1d_impl(oracle_lib):
    d = np.load(golden('units.npz'))
    for c, N in enumerate(d['pw_N']):
        for tf in (0, 1):
            tag = 'pw%d_%d' % (c, tf)
            I = np.zeros(N)
            Psi = np.zeros(N)
            oracle_lib.dll.lsx_oracle_piecewise_1d_impl(float(d[tag + '_mu']), tf, float(d[tag + '_Istart']), int(N),
                                                        _ptr(d[tag + '_z']), _ptr(d[tag + '_chi']), _ptr(d[tag + '_S']),
                                                        _ptr(I), _ptr(Psi))
            assert relerr(I, d[tag + '_I']) < 2e-12, tag
            assert relerr(Psi, d[tag + '_Psi'], floor=1e-300) < 2e-12 or np.allclose(Psi, d[tag + '_Psi'], rtol=2e-12, atol=1e-30), tag
            assert Psi[N - 1 if tf else 0] == 0.0


def test_piecewise_linear_1d_boundary_conditions(oracle_lib):
    d = np.load(golden('units.npz'))
    chi, S = d['pl_chi'], d['pl_S']
    rays = [(wi, mu, tf) for wi in range(4) for mu in range(2) for tf in (0, 1)]
    I, Psi = oracle_lib.piecewise_linear_1d(d['pl_height'], d['pl_temperature'],
                                            [d['pl_muz'][m] for _, m, _ in rays], [tf for *_, tf in rays],
                                            [d['pl_wav'][w] for w, _, _ in rays],
                                            np.tile(chi, (len(rays), 1)), np.tile(S, (len(rays), 1)))
    for r, (wi, mu, tf) in enumerate(rays):
        assert relerr(I[r], d['pl_I_%d_%d_%d' % (wi, mu, tf)], floor=1e-300) < 2e-12
        assert np.allclose(Psi[r], d['pl_Psi_%d_%d_%d' % (wi, mu, tf)], rtol=2e-12, atol=0)


# Tolerances: the reference's w2 (formal_solver.py:41-43) forms w1 = (1 - e) - dtau e, which
# cancels to ~dtau^2/2; a 1-ulp difference between numpy's SIMD exp and libm's exp is an ABSOLUTE
# 1e-16 error on w1, i.e. up to 1e-9 RELATIVE when dtau is just above the 5e-4 Taylor switch.  FALC
# Ca+H has such an interval on the Ly-beta core ray (dtau = 6.1e-4) which moves that ray's I by
# 8e-12; FALC CaII has none (2.4e-13).  So the single-call bar is 1e-12 (CaII) / 3e-11 (Ca+H).
@pytest.mark.parametrize('name,compact,tol', [('falc_ca.npz', True, 1e-12), ('falc_ca.npz', False, 1e-12),
                                              ('falc_cah.npz', True, 3e-11)])
def test_first_calls_match_reference(oracle_lib, name, compact, tol):
    """calls 1-4: the single-call bars above.  Behind the first statistical equilibrium the bars are COMPUTED (tests/envelope.py,
    SequenceBars: the oracle's own +-1-ulp-exp spread through the same calls + the LU's componentwise conditioning)"""
    prob, block, d = fixtures.load_problem_npz(golden(name), phi_compact=compact)

    def make():
        e = Engine(prob, 1, lib=oracle_lib)
        e.set_columns(0, block)
        return e
    bars = envelope.SequenceBars(oracle_lib, make, prob, 5, 3, tol)
    eng = make()
    dn = dn_prev = 0.0
    for it in range(1, 6):
        dJ = eng.formal_sol_gamma()
        tag = 'fs%d' % it
        if tag + '_dJ' in d:
            assert dJ == pytest.approx(float(d[tag + '_dJ']), rel=1e-9)
            assert relerr(eng.get(_capi.LSX_I)[0], d[tag + '_I']) < (tol if it < 5 else bars.I_bar(it - 1, dn))
            if tag + '_J' in d:
                if it < 5:
                    assert relerr(eng.get(_capi.LSX_J)[0], d[tag + '_J']) < tol
                else:
                    bars.check_J(eng.get(_capi.LSX_J), d[tag + '_J'][None], it - 1, dn, ' (oracle vs reference)')
            off, diag = gamma_err(eng.get(_capi.LSX_GAMMA)[0], fixtures.gamma_from_raw(d, tag, prob), prob)
            bo, bd = (10 * tol, tol) if it < 5 else bars.gamma_bar(it - 1, dn, gamma_err)
            assert off < bo and diag < bd, (it, off, diag, bo, bd)
        if it > 3:
            dP = eng.stat_equil()
            if 'se%d_dPops' % it in d:
                assert dP == pytest.approx(float(d['se%d_dPops' % it]), rel=1e-7)
                dn_prev, dn = dn, bars.check_n(eng.get(_capi.LSX_N), fixtures.pops_from_raw(d, 'se%d' % it, prob)[None], it - 1, ' (oracle vs reference)', dn)
    eng.close()


# The reference's larger model atoms (rh_atoms.py:194 C_atom, :355 Fe_simple_atom, :50 MgII_atom; tests/golden/make_golden.py,
# gen_falc_multilevel): 11-15 levels, 15-16 lines and 10-14 bound-free continua of ONE atom, up to 14 transitions at a wavelength,
# every continuum sharing its atom with the lines it overlaps -- the atom.eta / atom.chi / atom.U cross terms between lines and
# continua of rh_method.py:606-627, 654-681 that CaII (+ H) never reaches.  Measured oracle-vs-reference: I 3.4e-12 (C; the w2
# cancellation next to the Taylor switch, as for Ca+H), J 7e-13, Gamma off-diagonal 6e-14; populations after the first statistical
# equilibrium 1.1e-8 (Fe: a 15-level system, LU rounding x conditioning; dPops itself to 5e-10).  C's J holds zeros at its
# shortest wavelengths, so dJ = 1.0 on every call (rh_method.py:705-706) -- reproduced.
@pytest.mark.parametrize('name,compact', [('falc_c.npz', True), ('falc_fe.npz', True), ('falc_mg.npz', True), ('falc_mg.npz', False)])
def test_multilevel_reference_atoms_match_reference(oracle_lib, name, compact):
    prob, block, d = fixtures.load_problem_npz(golden(name), phi_compact=compact)
    assert prob.Natoms == 1 and prob.Nlevel[0] >= 11 and prob.Ntrans >= 25
    eng = Engine(prob, 1, lib=oracle_lib)
    eng.set_columns(0, block)
    tol = 1e-11
    for it in range(1, 5):
        dJ = eng.formal_sol_gamma()
        tag = 'fs%d' % it
        assert dJ == pytest.approx(float(d[tag + '_dJ']), rel=1e-9)
        assert relerr(eng.get(_capi.LSX_I)[0], d[tag + '_I']) < tol
        if tag + '_J' in d:
            assert relerr(eng.get(_capi.LSX_J)[0], d[tag + '_J'], floor=1e-300) < tol
        off, diag = gamma_err(eng.get(_capi.LSX_GAMMA)[0], fixtures.gamma_from_raw(d, tag, prob), prob)
        assert off < 1e-12 and diag < 1e-12, (it, off, diag)
    dP = eng.stat_equil()
    assert dP == pytest.approx(float(d['se4_dPops']), rel=1e-8)
    # computed bar (tests/envelope.py): carbon 1.3e-11, MgII 1.4e-8, iron 7e-8 -- three times what ONE rounding of a system with that
    # atom's componentwise condition number does, plus the exp envelope (round 5 had a flat 1e-7 for the three)

    def make():
        e = Engine(prob, 1, lib=oracle_lib)
        e.set_columns(0, block)
        return e
    bars = envelope.SequenceBars(oracle_lib, make, prob, 4, 3, tol)
    bars.check_n(eng.get(_capi.LSX_N), fixtures.pops_from_raw(d, 'se4', prob)[None], 3, ' (oracle vs reference)')
    eng.close()


# All five of the reference's model atoms in the set and active at once (round 5, tests/golden/make_golden.py gen_falc_all): 53
# levels, 109 transitions, 3966 wavelengths, up to 44 bound-free continua of five atoms at one wavelength (91 nm), five Gamma matrices
# and five statistical equilibria per iteration (rh_method.py:586-590, 710).  Measured oracle-vs-reference: I 8.4e-12 (the w2
# cancellation next to the Taylor switch), J 2.6e-12, Gamma 1e-13 off the diagonal; after the first statistical equilibrium dPops =
# 277.97 (the LTE start is far off for C and Fe) to 2e-10 and the populations to 1.4e-8.
def test_all_five_reference_atoms_active_match_reference(oracle_lib):
    prob, block, d = fixtures.load_problem_npz(golden('falc_all.npz'))
    assert prob.Natoms == 5 and sum(prob.Nlevel) == 53 and prob.Ntrans == 109 and prob.Nspect == 3966
    eng = Engine(prob, 1, lib=oracle_lib)
    eng.set_columns(0, block)
    for it in range(1, 5):
        dJ = eng.formal_sol_gamma()
        tag = 'fs%d' % it
        if tag + '_I' not in d:
            continue
        assert dJ == pytest.approx(float(d[tag + '_dJ']), rel=1e-9)
        assert relerr(eng.get(_capi.LSX_I)[0], d[tag + '_I']) < 2e-11
        if tag + '_J' in d:
            assert relerr(eng.get(_capi.LSX_J)[0], d[tag + '_J'], floor=1e-300) < 1e-11
        off, diag = gamma_err(eng.get(_capi.LSX_GAMMA)[0], fixtures.gamma_from_raw(d, tag, prob), prob)
        assert off < 1e-12 and diag < 1e-12, (it, off, diag)
    dP = eng.stat_equil()
    assert dP == pytest.approx(float(d['se4_dPops']), rel=1e-8)
    # per atom, computed (tests/envelope.py): hydrogen 2e-11, carbon 1.3e-11, MgII 2.6e-8, CaII 9.5e-10, iron 7.6e-8

    def make():
        e = Engine(prob, 1, lib=oracle_lib)
        e.set_columns(0, block)
        oracle_lib.dll.lsx_oracle_set_threads(e._h, 8)
        return e
    bars = envelope.SequenceBars(oracle_lib, make, prob, 4, 3, 2e-11)
    bars.check_n(eng.get(_capi.LSX_N), fixtures.pops_from_raw(d, 'se4', prob)[None], 3, ' (oracle vs reference)')
    eng.close()


def test_rates_quirk_accumulate_across_calls(oracle_lib):
    """rh_method.py:691-692: Rij/Rji are never zeroed and Rji uses Vij.  Not part of the ABI (nothing reads
    them in the reference); the oracle keeps them because the golden files hold them and they test I at every depth."""
    import ctypes as C
    f = oracle_lib.dll.lsx_oracle_rates
    f.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    prob, block, d = fixtures.load_problem_npz(golden('falc_ca.npz'))
    eng = Engine(prob, 1, lib=oracle_lib)
    eng.set_columns(0, block)
    for it in (1, 2):
        eng.formal_sol_gamma()
        Rij, Rji = np.empty((prob.Ntrans, prob.Nspace)), np.empty((prob.Ntrans, prob.Nspace))
        assert f(eng._h, 0, Rij.ctypes.data_as(C.POINTER(C.c_double)), Rji.ctypes.data_as(C.POINTER(C.c_double))) == 0
        for kr in range(prob.Ntrans):
            assert relerr(Rij[kr], d['fs%d_Rij_t%d' % (it, kr)]) < 1e-11
            assert relerr(Rji[kr], d['fs%d_Rji_t%d' % (it, kr)]) < 1e-11


def test_falc_ca_trajectory_and_converged_state(oracle_lib):
    """D1 (test.py:20-29): 46 iterations, same dJ/dPops trajectory, same converged n, J, I."""
    prob, block, d = fixtures.load_problem_npz(golden('falc_ca.npz'))
    eng = Engine(prob, 1, lib=oracle_lib)
    eng.set_columns(0, block)
    h = drivers.iterate_mali(EngineAdapter(eng))
    assert h.converged and h.n_iter == int(d['n_iter']) == 46
    assert np.allclose(h.dJ, d['traj_dJ'], rtol=1e-6)
    assert np.allclose(h.dPops[3:], d['traj_dPops'][3:], rtol=1e-6)
    assert relerr(eng.get(_capi.LSX_N)[0], fixtures.pops_from_raw(d, 'conv', prob)) < 1e-7
    assert relerr(eng.get(_capi.LSX_J)[0], d['conv_J']) < 1e-7
    assert relerr(eng.get(_capi.LSX_I)[0], d['conv_I']) < 1e-7
    # SURVEY 8c anchor values
    I = eng.get(_capi.LSX_I)[0]
    la = int(np.argmin(np.abs(prob.wavelength - 500.0)))
    assert I[la, 4] == pytest.approx(3.4535686980e-08, rel=1e-9)


def test_nonzero_vlos_four_dimensional_profile(oracle_lib):
    """vlos != 0: phi depends on (mu, direction) (rh_method.py:229-240).  The fixture holds the profile's inputs and a
    strided sample of the reference's phi: the profiles are built through lsx_set_line_profiles and checked on it."""
    prob, block, d = fixtures.load_problem_npz(golden('falc_ca_vlos.npz'))
    assert not prob.phi_compact and block.phi is None
    eng = Engine(prob, 1, lib=oracle_lib)
    eng.set_columns(0, block)
    eng.set_line_profiles(0, *fixtures.profile_inputs(prob, d))
    phi, wphi = eng.get(_capi.LSX_PHI)[0], eng.get(_capi.LSX_WPHI)[0]
    o = 0
    for kr, t in enumerate(prob.trans):
        if t.is_line:
            assert relerr(phi[o:o + t.Nlambda][::7, :, :, ::9], d['t%d_phi_sample' % kr]) < 1e-13
            assert relerr(wphi[prob.lines.index(t)], d['t%d_wphi' % kr]) < 1e-13
            o += t.Nlambda
    for it in range(1, 6):
        dJ = eng.formal_sol_gamma()
        tag = 'fs%d' % it
        if tag + '_dJ' in d:
            assert dJ == pytest.approx(float(d[tag + '_dJ']), rel=1e-9)
            assert relerr(eng.get(_capi.LSX_I)[0], d[tag + '_I']) < (1e-12 if it < 5 else 1e-9)
            assert relerr(eng.get(_capi.LSX_J)[0], d[tag + '_J']) < (1e-12 if it < 5 else 1e-9)
        if it > 3:
            eng.stat_equil()


def test_error_conventions(oracle_lib):
    prob, block, d = fixtures.load_problem_npz(golden('falc_ca.npz'))
    eng = Engine(prob, 1, lib=oracle_lib)
    with pytest.raises(_capi.LsxError):
        eng.get(_capi.LSX_J, col0=1, ncol=1)      # out of range
    with pytest.raises(ValueError):
        bad = block.slice(0, 1)
        bad.height = bad.height[:, :-1]
        eng.set_columns(0, bad)
    # singular Gamma -> LinAlgError, like scipy.linalg.solve at rh_method.py:739
    eng.set_columns(0, block)
    with pytest.raises(np.linalg.LinAlgError):
        eng.stat_equil()   # Gamma is all zeros before any formal solution
