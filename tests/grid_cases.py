"""Bodies of the wavelength-grid / active-set tests, shared by the CPU run (oracle) and the GPU-box run
(the product's host code inside liblsx_hip.so).  Pinned on tests/golden/setup_falc.npz: the reference's own merged grid,
blueIdx, per-transition Nlambda and active sets for FALC H + CaII, and its lines' and continua's local grids."""
import numpy as np
import pytest

from conftest import golden
from lightspinner_amd import _capi


def _falc_grids(d):
    grids, is_line, edges = [], [], []
    for a, il, q in d['grid_trans_order']:
        kind = 'line' if il else 'cont'
        grids.append(d['m%d_%s%d_grid0' % (a, kind, q)])
        is_line.append(bool(il))
        edges.append(0.0 if il else float(d['m%d_cont_edge' % a][q]))
    return grids, is_line, edges


def reference_grid_bit_exact(lib):
    d = dict(np.load(golden('setup_falc.npz')))
    grids, is_line, edges = _falc_grids(d)
    wav, blue, red = lib.wavelength_grid(grids, is_line, edges, lambda_reference=float(d['grid_lambdaReference']))
    assert np.array_equal(wav, d['grid_wavelength'])                       # bit exact: sort / unique of the same doubles
    assert np.array_equal(blue, d['grid_blueIdx']) and np.array_equal(red - blue, d['grid_Nlambda'])
    assert np.array_equal(lib.active_set(blue, red, wav.shape[0]), d['grid_active'])
    for kr, (a, il, q) in enumerate(d['grid_trans_order']):                # the transitions' grids after the merge
        if not il:
            assert np.array_equal(wav[blue[kr]:red[kr]], d['m%d_cont%d_wavelength' % (a, q)])
    # the tables the hot path is given (falc_cah.npz) are this grid; active rows in the problem's transition order
    p = dict(np.load(golden('falc_cah.npz')))
    assert np.array_equal(wav, p['wavelength'])
    key = {(int(a), bool(il), int(i), int(j)): kr for kr, (a, il, i, j) in enumerate(zip(p['t_atom'], p['t_isline'], p['t_i'], p['t_j']))}
    for kr, (a, il, q) in enumerate(d['grid_trans_order']):
        pre = 'm%d_%s_' % (a, 'line' if il else 'cont')
        row = key[(int(a), bool(il), int(d[pre + 'i'][q]), int(d[pre + 'j'][q]))]
        assert blue[kr] == p['t_Nblue'][row] and red[kr] - blue[kr] == p['t_Nlambda'][row]
        assert np.array_equal(lib.active_set(blue, red, wav.shape[0])[kr], p['t_active'][row])


def line_grids_and_continuum_alpha(lib):
    d = dict(np.load(golden('setup_falc.npz')))
    for a in range(len(d['atom_names'])):
        pre = 'm%d_' % a
        for q in range(d[pre + 'line_i'].shape[0]):
            w = lib.line_wavelength(d[pre + 'line_lambda0'][q], d[pre + 'line_qCore'][q], d[pre + 'line_qWing'][q], d[pre + 'line_NlambdaGen'][q])
            ref = d[pre + 'line%d_grid0' % q]
            assert w.shape == ref.shape and np.allclose(w, ref, rtol=4e-16, atol=0)      # exp / log / sqrt of two libms
            assert w[w.shape[0] // 2] == d[pre + 'line_lambda0'][q]
        for q in range(d[pre + 'cont_i'].shape[0]):
            kw = dict(edge=d[pre + 'cont_edge'][q], min_lambda=d[pre + 'cont_minLambda'][q])
            if d[pre + 'cont_hydrogenic'][q]:
                i, j = d[pre + 'cont_i'][q], d[pre + 'cont_j'][q]
                kw.update(alpha0=d[pre + 'cont_alpha0'][q], E_i=d[pre + 'lev_E_SI'][i], E_j=d[pre + 'lev_E_SI'][j], stage_j=d[pre + 'lev_stage'][j])
            else:
                kw.update(table=(d[pre + 'cont%d_grid0' % q], d[pre + 'cont%d_alpha_grid0' % q]))
            ref = d[pre + 'cont%d_alpha' % q]
            got = lib.continuum_alpha(d[pre + 'cont%d_wavelength' % q], **kw)
            assert np.max(np.abs(got - ref)) <= 1e-14 * np.max(ref), (a, q)             # measured 2.3e-15 (cubic), 1.5e-16 (Gaunt)
            # the local grid's own alpha (atomic_model.py:660 for hydrogenic continua, the table itself otherwise)
            x0 = d[pre + 'cont%d_grid0' % q]
            own = lib.continuum_alpha(x0, **kw)
            ref0 = np.where(x0 <= kw['edge'], d[pre + 'cont%d_alpha_grid0' % q], 0.0)    # some tables run past the edge: cut, :608
            assert np.max(np.abs(own - ref0)) <= 1e-14 * np.max(ref0)


def _numpy_merge(grids, is_line, edges, extra, lam_ref):
    """atomic_set.py:381-416 with numpy, for random inputs"""
    parts = ([np.asarray(extra)] if extra is not None else []) + [np.array([lam_ref])]
    for g, l, e in zip(grids, is_line, edges):
        parts += [g] if l else [np.array([e]), g[g <= e]]
    grid = np.unique(np.sort(np.concatenate(parts)))
    blue = [int(np.searchsorted(grid, g[0])) for g in grids]
    red = [int(np.searchsorted(grid, g[-1])) + 1 for g in grids]
    for kr, (l, e) in enumerate(zip(is_line, edges)):
        if not l:
            red[kr] = min(red[kr], grid.shape[0])      # a table ending beyond every other grid: the reference indexes out of
            while grid[red[kr] - 1] > e:               # bounds here (IndexError); the library clamps
                red[kr] -= 1
    return grid, np.array(blue), np.array(red)


def random_and_edge_cases(lib):
    rng = np.random.default_rng(11)
    for trial in range(20):
        grids, is_line, edges = [], [], []
        for kr in range(int(rng.integers(1, 9))):
            l = bool(rng.integers(0, 2))
            n = int(rng.integers(1 if l else 2, 40))
            g = np.sort(rng.choice(np.round(rng.uniform(50, 900, 60), 1), size=n))       # coarse values: duplicates across grids
            grids.append(g)
            is_line.append(l)
            # continua: the edge inside, at the end of, or beyond the tabulated grid (the grid may run past its edge, :411)
            edges.append(0.0 if l else float(rng.choice([g[-1], g[max(0, n // 2)] + 0.05, g[-1] + 3.0])))
        extra = np.round(rng.uniform(10, 1000, int(rng.integers(0, 5))), 1) if trial % 2 else None
        wav, blue, red = lib.wavelength_grid(grids, is_line, edges, extra=extra, lambda_reference=500.0)
        rw, rb, rr = _numpy_merge(grids, is_line, edges, extra, 500.0)
        assert np.array_equal(wav, rw) and np.array_equal(blue, rb) and np.array_equal(red, rr), trial
        act = lib.active_set(blue, red, wav.shape[0])
        assert np.array_equal(act, (np.arange(wav.shape[0])[None] >= rb[:, None]) & (np.arange(wav.shape[0])[None] < rr[:, None]))
    # no transitions: the reference wavelength alone
    wav, blue, red = lib.wavelength_grid([], [], [])
    assert np.array_equal(wav, [500.0]) and blue.shape == (0,)
    assert lib.active_set(blue, red, 1).shape == (0, 1)
    # errors: descending grid, range outside the grid
    with pytest.raises(_capi.LsxError):
        lib.wavelength_grid([np.array([3.0, 2.0])], [True], [0.0])
    with pytest.raises(_capi.LsxError):
        lib.active_set([0], [5], 4)
    with pytest.raises(_capi.LsxError):
        lib.continuum_alpha([100.0], edge=200.0, min_lambda=50.0, table=([50.0, 100.0, 200.0], [1.0, 2.0, 3.0]))   # < 4 points
    # an explicit continuum whose cubic undershoots falls back to linear interpolation everywhere (atomic_model.py:610-611)
    x = np.array([50.0, 60.0, 70.0, 80.0, 90.0, 100.0])
    y = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0])
    w = np.linspace(50.0, 100.0, 41)
    assert np.allclose(lib.continuum_alpha(w, edge=100.0, min_lambda=50.0, table=(x, y)), np.interp(w, x, y), rtol=1e-15, atol=1e-16)


def spectrum_configuration_feeds_context(lib, ctx_lib):
    """compute_wavelength_grid on model objects holding their LOCAL grids -> SpectrumConfiguration -> Context: the chain the
    reference runs (test.py:12-18) with the library doing the work; results against the reference's golden vectors"""
    from helpers import build_data_fakes
    from lightspinner_amd.spectrum import compute_wavelength_grid
    from lightspinner_amd.rh_method import Context
    from conftest import relerr
    d = dict(np.load(golden('falc_cah.npz')))
    s = dict(np.load(golden('setup_falc.npz')))
    atmos, spect0, eq, bg = build_data_fakes(d, s)
    models = spect0.radSet.activeAtoms
    names = [str(x) for x in s['atom_names']]
    for m in models:                                       # back to the state before the merge: local grids, tabulated alpha
        a = names.index(m.name)
        pos = {(int(i), int(j)): q for q, (i, j) in enumerate(zip(s['m%d_line_i' % a], s['m%d_line_j' % a]))}
        for l in m.lines:
            l.wavelength = s['m%d_line%d_grid0' % (a, pos[(l.i, l.j)])].copy()
        cpos = {(int(i), int(j)): q for q, (i, j) in enumerate(zip(s['m%d_cont_i' % a], s['m%d_cont_j' % a]))}
        for c in m.continua:
            q = cpos[(c.i, c.j)]
            c.wavelength, c.alpha = s['m%d_cont%d_grid0' % (a, q)].copy(), s['m%d_cont%d_alpha_grid0' % (a, q)].copy()
            c.lambdaEdge = float(s['m%d_cont_edge' % a][q])
            if s['m%d_cont_hydrogenic' % a][q]:
                c.alpha0, c.minLambda = float(s['m%d_cont_alpha0' % a][q]), float(s['m%d_cont_minLambda' % a][q])
                lev = lambda k: type('L', (), dict(E_SI=float(s['m%d_lev_E_SI' % a][k]), stage=int(s['m%d_lev_stage' % a][k])))
                c.iLevel, c.jLevel = lev(c.i), lev(c.j)
    spect = compute_wavelength_grid(models, lib=lib)
    assert np.array_equal(spect.wavelength, d['wavelength'])
    kr = 0
    for m in models:
        for t in m.lines + m.continua:
            assert spect.blueIdx[kr] == d['t_Nblue'][kr] and t.wavelength.shape[0] == d['t_Nlambda'][kr]
            assert np.array_equal(spect.active[kr], d['t_active'][kr])
            if not d['t_isline'][kr]:
                assert np.max(np.abs(t.alpha - d['t%d_alpha' % kr])) <= 1e-14 * np.max(d['t%d_alpha' % kr])
            kr += 1
    ctx = Context(atmos, spect, eq, bg, lib=ctx_lib)
    assert ctx.setup == 'native' and ctx.problem.Nspect == 777
    for it in (1, 2):
        dJ = ctx.formal_sol_gamma_matrices()
        assert dJ == pytest.approx(float(d['fs%d_dJ' % it]), rel=1e-8)
        assert relerr(ctx.I, d['fs%d_I' % it]) < 1e-10
