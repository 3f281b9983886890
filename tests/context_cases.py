"""Bodies of the drop-in boundary tests (rh_method.Context / formal_solver.piecewise_linear_1d of lightspinner_amd),
shared by the CPU run (oracle bound as the checker, tests/test_host_logic.py) and the GPU run (the product library,
tests/test_context_hip.py).  `lib=None` means the product default: the HIP library."""
import numpy as np
import pytest

from conftest import golden, relerr, gamma_err
from helpers import build_fakes, build_data_fakes
from lightspinner_amd.rh_method import Context


_ONE_ATOM_6 = type('P', (), dict(Natoms=1, Nlevel=[6], lev2_off=[0], Nspace=82))


def context_dropin_matches_reference_golden(lib):
    d = dict(np.load(golden('falc_ca.npz')))
    atmos, spect, eq, bg = build_fakes(d)
    ctx = Context(atmos, spect, eq, bg, lib=lib)
    assert atmos.nondim_calls == 1                                   # rh_method.py:553
    assert ctx.problem.phi_compact and not ctx.problem.sca_per_lambda
    atom = ctx.activeAtoms[0]
    assert atom.n is eq['CA'].pops                                   # aliasing contract, rh_method.py:412-416
    assert relerr(atom.trans[0].phi[:, 0, 0, :], d['t0_phi']) < 1e-13 and relerr(atom.trans[0].wphi, d['t0_wphi']) < 1e-13
    for it in range(1, 6):
        dJ = ctx.formal_sol_gamma_matrices()
        assert dJ == pytest.approx(float(d['fs%d_dJ' % it]), rel=1e-8)
        assert relerr(ctx.I, d['fs%d_I' % it]) < (1e-12 if it < 5 else 1e-8)
        assert relerr(ctx.J, d['fs%d_J' % it]) < (1e-12 if it < 5 else 1e-8)
        assert atom.Gamma.shape == d['fs%d_Gamma_a0' % it].shape
        off, diag = gamma_err(atom.Gamma.reshape(-1, 82), d['fs%d_Gamma_a0' % it].reshape(-1, 82), _ONE_ATOM_6)
        assert off < (1e-11 if it < 5 else 1e-7) and diag < (1e-12 if it < 5 else 1e-8), (it, off, diag)
        if it > 3:
            n_before = atom.n
            dP = ctx.stat_equil()
            assert dP == pytest.approx(float(d['se%d_dPops' % it]), rel=1e-7)
            assert atom.n is n_before and eq['CA'].n is atom.n       # updated in place
            assert relerr(atom.n, d['se%d_n_a0' % it]) < 1e-7
    with pytest.raises(AttributeError):
        atom.trans[0].Rij


def context_warm_start_and_host_edits(lib):
    d = dict(np.load(golden('falc_ca.npz')))
    atmos, spect, eq, bg = build_fakes(d, start_pops=[d['conv_n_a0']])
    ctx = Context(atmos, spect, eq, bg, lib=lib)
    assert np.array_equal(ctx.activeAtoms[0].n, d['conv_n_a0'])      # response_fn.py:33
    ctx.J[...] = d['conv_J']                                         # caller edits are honoured
    dJ = ctx.formal_sol_gamma_matrices()
    assert dJ < 2e-3
    ctx.activeAtoms[0].n[...] = d['a0_nStar']                        # back to LTE in place
    ctx.formal_sol_gamma_matrices()
    dP = ctx.stat_equil()
    assert dP > 0.1


def context_lazy_readback_keeps_the_reference_semantics(lib):
    """round 6: J, I and Gamma are fetched when they are first looked at (test.py:20-29 and response_fn.py:11-21 read them after
    the loop) -- and from then on rewritten IN PLACE by every call, like the reference's live arrays (rh_method.py:562-563, 640, 638,
    587-590), with edits of J honoured.  A driver that never looks pays no read-back; one that holds the arrays sees what the
    reference would show it.  atom.n (the caller's own array, :412-416) is written back by every stat_equil either way."""
    d = dict(np.load(golden('falc_ca.npz')))
    atmos, spect, eq, bg = build_fakes(d)
    ctx = Context(atmos, spect, eq, bg, lib=lib)
    fetched = []
    real_get = ctx._engine.get
    ctx._engine.get = lambda what, *a, **k: (fetched.append(what), real_get(what, *a, **k))[1]
    atom = ctx.activeAtoms[0]
    for it in range(1, 4):
        ctx.formal_sol_gamma_matrices()
    assert fetched == []                                             # three formal solutions, nothing read back
    assert relerr(ctx.J, d['fs3_J']) < 1e-12                         # first look: fetched now
    assert len(fetched) == 1
    Jheld, Iheld = ctx.J, ctx.I
    assert Jheld is ctx.J and Iheld is ctx.I                         # one array object each for the life of the context
    assert relerr(Iheld, d['fs3_I']) < 1e-12
    G = atom.Gamma
    assert gamma_err(G.reshape(-1, 82), d['fs3_Gamma_a0'].reshape(-1, 82), _ONE_ATOM_6)[0] < 1e-11
    ctx.formal_sol_gamma_matrices()                                  # the held arrays follow, in place
    assert relerr(Jheld, d['fs4_J']) < 1e-12 and relerr(Iheld, d['fs4_I']) < 1e-12
    assert G is atom.Gamma and gamma_err(G.reshape(-1, 82), d['fs4_Gamma_a0'].reshape(-1, 82), _ONE_ATOM_6)[0] < 1e-11
    n_before = atom.n
    dP = ctx.stat_equil()
    assert dP == pytest.approx(float(d['se4_dPops']), rel=1e-7)
    assert atom.n is n_before and eq['CA'].n is atom.n and relerr(atom.n, d['se4_n_a0']) < 1e-7
    Jheld[...] = d['conv_J']                                         # an edit through the held reference reaches the device
    atom.n[...] = d['conv_n_a0']
    assert ctx.formal_sol_gamma_matrices() < 2e-3
    ctx._engine.get = real_get
    # readback='eager' is round 5's behaviour from the first call on
    atmos, spect, eq, bg = build_fakes(d)
    ctx2 = Context(atmos, spect, eq, bg, lib=lib, readback='eager')
    ctx2.formal_sol_gamma_matrices()
    assert relerr(ctx2._host['J'], d['fs1_J']) < 1e-12 and not ctx2._stale['J']
    ctx.close(); ctx2.close()


def context_lookahead_gives_the_plain_sequence_bit_for_bit(lib):
    """round 6: stat_equil() enqueues the next formal solution ahead (lsx_sync_begin_populations / lsx_formal_sol_gamma_speculative);
    a driver that does anything but call formal_sol_gamma_matrices() next gets the speculative call taken back.  The same script of
    calls, looks and edits with and without look-ahead: every number identical."""
    d = dict(np.load(golden('falc_ca.npz')))

    def script(lookahead):
        atmos, spect, eq, bg = build_fakes(d)
        ctx = Context(atmos, spect, eq, bg, lib=lib, lookahead=lookahead)
        atom = ctx.activeAtoms[0]
        out = []
        for it in range(1, 7):                                       # test.py:20-29
            out.append(ctx.formal_sol_gamma_matrices())
            if it > 3:
                out.append(ctx.stat_equil())
                assert ctx._spec == bool(lookahead)                  # the next formal solution is on its way
                out.append(atom.n.copy())                            # written back already: rh_method.py:412-416
        out.append(ctx.stat_equil())                                 # a second stat_equil in a row: the same Gamma (taken back first)
        out.append(atom.n.copy())
        out.append(ctx.I.copy()); out.append(ctx.J.copy()); out.append(atom.Gamma.copy())      # a look: the last ACCEPTED call's results
        assert not ctx._spec
        out.append(ctx.formal_sol_gamma_matrices())
        out.append(ctx.stat_equil())
        atom.n[...] = d['a0_nStar']                                  # an edit behind a stat_equil: the speculative call used the old n
        out.append(ctx.formal_sol_gamma_matrices())
        out.append(ctx.I.copy())
        out.append(ctx.stat_equil())
        ctx.J[...] = d['conv_J']                                     # ... and an edit of J
        out.append(ctx.formal_sol_gamma_matrices())
        out.append(ctx.J.copy())
        out.append(ctx.stat_equil())
        ctx.update_collisions()
        out.append(ctx.formal_sol_gamma_matrices())
        out.append(ctx.I.copy())
        ctx.close()
        return out
    a, b = script(False), script(True)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))


def context_two_active_atoms_order_and_shapes(lib):
    d = dict(np.load(golden('falc_cah.npz')))
    atmos, spect, eq, bg = build_fakes(d)
    ctx = Context(atmos, spect, eq, bg, lib=lib)
    assert [a.atomicModel.name for a in ctx.activeAtoms] == ['H', 'CA']   # ascending atomic weight
    assert ctx.problem.Ntrans == 25 and ctx.problem.Nspect == 777
    dJ = ctx.formal_sol_gamma_matrices()
    assert dJ == 1.0
    assert relerr(ctx.I, d['fs1_I']) < 3e-11
    for a in range(2):
        off, diag = gamma_err(ctx.activeAtoms[a].Gamma.reshape(-1, 82), d['fs1_Gamma_a%d' % a].reshape(-1, 82),
                              _ONE_ATOM_6)
        assert off < 3e-10 and diag < 3e-11


def context_native_setup_chain(lib, name='falc_cah.npz'):
    """models that carry their atomic data: Context hands them to lsx_set_atomic_data and the library derives vBroad,
    aDamp, the profiles and the collisional rates from the atmosphere (lsx_set_atmosphere); everything the reference's
    constructor would have computed is read back and compared with what the reference did compute (1e-13: the device
    evaluates the same formulas with its own exp / log / sqrt / Faddeeva; phi on a ray-dependent profile 1e-12)"""
    d = dict(np.load(golden(name)))
    s = dict(np.load(golden('setup_falc.npz')))
    atmos, spect, eq, bg = build_data_fakes(d, s)
    ctx = Context(atmos, spect, eq, bg, lib=lib)
    assert ctx.setup == 'native'
    kr = 0
    for a, atom in enumerate(ctx.activeAtoms):
        assert relerr(atom.vBroad, d['a%d_vBroad' % a]) < 1e-14
        C, Cref = atom.C, d['a%d_C' % a]
        assert C.shape == Cref.shape
        assert np.max(np.abs(C - Cref)) <= 1e-12 * np.max(np.abs(Cref))
        for t in atom.trans:
            if t.isLine:
                assert relerr(t.aDamp, d['t%d_aDamp' % kr]) < 1e-13
                assert relerr(t.wphi, d['t%d_wphi' % kr]) < 1e-12
                if ('t%d_phi' % kr) in d:
                    ref = d['t%d_phi' % kr]
                    got = t.phi if ref.ndim == 4 else t.phi[:, 0, 0, :]
                    assert t.phi.shape == (t.wavelength.shape[0], atmos.Nrays, 2, atmos.Nspace)
                    assert relerr(got, ref) < 1e-12
                elif ('t%d_phi_sample' % kr) in d:
                    assert relerr(t.phi[::7, :, :, ::9], d['t%d_phi_sample' % kr]) < 1e-12
            kr += 1
    tol = 1e-10 if name == 'falc_cah.npz' else 1e-11     # set-up differences (1e-13) propagate through Gamma and J
    for it in range(1, 5):
        dJ = ctx.formal_sol_gamma_matrices()
        if 'fs%d_dJ' % it in d:
            assert dJ == pytest.approx(float(d['fs%d_dJ' % it]), rel=1e-8)
            assert relerr(ctx.I, d['fs%d_I' % it]) < tol
            if 'fs%d_J' % it in d:
                assert relerr(ctx.J, d['fs%d_J' % it]) < tol
    dP = ctx.stat_equil()
    if 'se4_dPops' in d:
        assert dP == pytest.approx(float(d['se4_dPops']), rel=1e-6)
    J = ctx.J.copy()
    ctx.update_collisions()                              # re-derivation leaves the state where it was
    assert np.max(np.abs(ctx.activeAtoms[0].C - d['a0_C'])) <= 1e-12 * np.max(np.abs(d['a0_C']))
    assert np.array_equal(ctx._engine.get(1)[0], J)   # LSX_J
    ctx.close()


def context_methods_setup_is_still_the_reference_interface(lib):
    """models that only offer v_broad / damping / compute_rates (the reference's own interface): those are called and the
    profiles are built by the library from their results"""
    d = dict(np.load(golden('falc_ca_vlos.npz')))
    atmos, spect, eq, bg = build_fakes(d)
    ctx = Context(atmos, spect, eq, bg, lib=lib)
    assert ctx.setup == 'methods' and not ctx.problem.phi_compact
    for kr, t in enumerate(ctx.activeAtoms[0].trans):
        if t.isLine:
            assert relerr(t.phi[::7, :, :, ::9], d['t%d_phi_sample' % kr]) < 1e-13
            assert relerr(t.wphi, d['t%d_wphi' % kr]) < 1e-13
            assert np.array_equal(t.aDamp, d['t%d_aDamp' % kr])
    assert np.array_equal(ctx.activeAtoms[0].vBroad, d['a0_vBroad'])
    assert np.array_equal(ctx.activeAtoms[0].C, d['a0_C'])
    assert ctx.formal_sol_gamma_matrices() == pytest.approx(float(d['fs1_dJ']), rel=1e-9)
    assert relerr(ctx.I, d['fs1_I']) < 1e-12
    with pytest.raises(ValueError):
        Context(*build_fakes(d), lib=lib, setup='other')


class _Atmos:
    """what formal_solver.piecewise_linear_1d reads from an Atmosphere (formal_solver.py:144-212)"""
    def __init__(self, d):
        self.height, self.temperature, self.muz = d['pl_height'], d['pl_temperature'], d['pl_muz']
        self.Nspace = self.height.shape[0]


def piecewise_linear_1d_dropin(lib):
    """lightspinner_amd.formal_solver.piecewise_linear_1d(atmos, mu, toFrom, wav, chi, S) -> IPsi, the reference's own
    signature (formal_solver.py:144), against the reference's outputs for 16 rays (boundary conditions of both
    directions, formal_solver.py:203-209)"""
    from lightspinner_amd.formal_solver import piecewise_linear_1d, IPsi
    d = np.load(golden('units.npz'))
    atmos = _Atmos(d)
    for wi in range(4):
        for mu in range(2):
            for tf in (False, True):
                r = piecewise_linear_1d(atmos, mu, tf, float(d['pl_wav'][wi]), d['pl_chi'], d['pl_S'], lib=lib)
                assert isinstance(r, IPsi) and r.I.shape == r.PsiStar.shape == (atmos.Nspace,)
                tag = '%d_%d_%d' % (wi, mu, int(tf))
                assert relerr(r.I, d['pl_I_' + tag], floor=1e-300) < 5e-12
                assert np.allclose(r.PsiStar, d['pl_Psi_' + tag], rtol=5e-12, atol=0)
                assert r.PsiStar[-1 if tf else 0] == 0.0


def golden_w2_and_piecewise_1d_impl(lib):
    """the reference's unit vectors for w2 (407 optical depths over the three branches and both thresholds,
    formal_solver.py:14-44) and piecewise_1d_impl (10 grids, N = 3 ... 200, both directions, formal_solver.py:46-142)
    through the C ABI (lsx_w2 / lsx_piecewise_1d_impl)"""
    d = np.load(golden('units.npz'))
    w = lib.w2(d['w2_dtau'])
    # exp() implementations differ by an ulp of exp(-dtau) <= 1; w0 = 1 - e and w1 = w0 - dtau e inherit that as an
    # ABSOLUTE error (the formulas cancel) -- same bar as the oracle's own test
    assert np.allclose(w, d['w2_out'], rtol=4e-16, atol=5e-16)
    small, large = d['w2_dtau'] < 5e-4, d['w2_dtau'] > 50.0
    assert small.any() and large.any() and (~small & ~large).any()
    assert np.array_equal(w[large], np.ones((int(large.sum()), 2)))          # saturated branch is exact
    assert np.allclose(w[small], d['w2_out'][small], rtol=3e-16, atol=0)      # Taylor branch: no exp involved
    for c, N in enumerate(d['pw_N']):
        for tf in (0, 1):
            tag = 'pw%d_%d' % (c, tf)
            I, Psi = lib.piecewise_1d_impl(d[tag + '_z'], [float(d[tag + '_mu'])], [tf], [float(d[tag + '_Istart'])],
                                           d[tag + '_chi'][None], d[tag + '_S'][None])
            assert relerr(I[0], d[tag + '_I']) < 2e-12, tag
            assert np.allclose(Psi[0], d[tag + '_Psi'], rtol=2e-12, atol=1e-30), tag
            assert Psi[0][N - 1 if tf else 0] == 0.0
    # many rays of one grid in one call (ragged batch sizes around the 64-lane launch width)
    tag = 'pw4_1'
    for nray in (1, 63, 64, 65, 130):
        I, Psi = lib.piecewise_1d_impl(d[tag + '_z'], [float(d[tag + '_mu'])] * nray, [1] * nray, [float(d[tag + '_Istart'])] * nray,
                                       np.tile(d[tag + '_chi'], (nray, 1)), np.tile(d[tag + '_S'], (nray, 1)))
        assert np.array_equal(I, np.tile(I[0], (nray, 1))) and relerr(I[-1], d[tag + '_I']) < 2e-12
    # empty input
    I, Psi = lib.piecewise_1d_impl(d[tag + '_z'], [], [], [], np.zeros((0, 82)), np.zeros((0, 82)))
    assert I.shape == (0, 82)
    assert lib.w2([]).shape == (0, 2)


def dead_level_nan_is_dropped_from_dpops(lib):
    """A level that nothing populates: the first stat_equil drives it to exactly 0 (relative change inf, which the
    reference reports: abs(1 - nOld/0)), the second sees 0/0 = NaN for it.  rh_method.py:741 takes the depth's
    change.max() (NaN, numpy) into Python's builtin max(maxRelChange, .), which keeps maxRelChange: that depth drops
    out and the call returns the largest finite change of the other depths / atoms -- never NaN."""
    from toy import toy_problem
    from lightspinner_amd.problem import Engine
    from lightspinner_amd import _capi, drivers
    prob, block = toy_problem(seed=11, ncol=3, dead_level=True)
    e = Engine(prob, block.ncol, lib=lib)
    e.set_columns(0, block)
    for _ in range(3):
        e.formal_sol_gamma()
    assert e.stat_equil() == float('inf')
    n = e.get(_capi.LSX_N)
    assert np.all(n[:, prob.NLtot - 1, :] == 0.0)
    e.formal_sol_gamma()
    dP = e.stat_equil()
    assert np.isfinite(dP) and dP > 0.0
    per_col = e.get(_capi.LSX_DPOPS_COL)
    assert np.all(np.isfinite(per_col)) and per_col.max() == dP
    # the level stays at 0 and the other levels keep converging
    e.formal_sol_gamma()
    assert e.stat_equil() < dP
    assert np.all(e.get(_capi.LSX_N)[:, prob.NLtot - 1, :] == 0.0)
    # per-column driver: a column stops by the reference's own comparisons; none is kept alive by a NaN
    rep = {}
    it = drivers.iterate_mali_columns(e, max_iter=60, report=rep)
    assert it.max() < 60
    e.close()
    return dP
