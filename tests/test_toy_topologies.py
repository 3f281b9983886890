"""HIP vs oracle on made-up atoms with awkward topologies (tests/toy.py): chained continua (generic fast
path), lines sharing levels, continuum-only atoms, 1 / 3 / 5 rays, odd depths, per-wavelength scattering.

Tolerances: one formal-solution call 1e-11 relative on I, J and off-diagonal Gamma (the two sides evaluate
exp() with different libraries); after 8 MALI iterations 1e-8 on the populations."""
import numpy as np
import pytest

import envelope
from conftest import relerr, gamma_err
from toy import toy_problem
from lightspinner_amd import _capi
from lightspinner_amd.problem import Engine

CASES = [
    dict(seed=1),                                                     # 3 rays, 37 depths, chained continua
    dict(seed=2, Nrays=1, Nspace=20, Nspect=70),                      # 64 wavelengths per wavefront
    dict(seed=3, Nrays=5, Nspace=82, Nspect=130, sca_per_lambda=True),
    dict(seed=4, phi_compact=True, chain=False),
    dict(seed=5, Nrays=2, Nspace=5, Nspect=40, ncol=40),              # shortest useful column; >= 32 columns: per-class launches
    dict(seed=6, Nrays=7, Nspace=33, Nspect=55, ncol=33),
    dict(seed=7, Nrays=8, Nspace=21, Nspect=48, ncol=2),              # 8 wavelengths per wavefront, all 64 lanes used
    dict(seed=8, Nrays=3, Nspace=700, Nspect=30, ncol=33),            # deep column: the operand table no longer fits LDS -> generic instance
    dict(seed=9, Nrays=3, Nspace=700, Nspect=30, ncol=2),             # same through the fused small-batch launch
    # regular bound-free sets on even tile widths: the column-mapped epilogue (k_fast_gamma_cols), with linked continua
    dict(seed=10, Nrays=4, Nspace=45, Nspect=100, ncol=34, chain=False),
    dict(seed=11, Nrays=5, Nspace=82, Nspect=140, ncol=3, chain=False),
    dict(seed=12, Nrays=8, Nspace=31, Nspect=64, ncol=33, chain=False, phi_compact=True),
    # multiplets under linked continua: three per-ray slots (the <3, 3, linked> instance) and four (no such instance: the plan
    # files the tile under the generic linked kernel), through the per-class launches (>= 32 columns) and the fused launch
    dict(seed=13, Nrays=5, Nspace=41, Nspect=120, ncol=34, multiplet=3),
    dict(seed=14, Nrays=5, Nspace=41, Nspect=120, ncol=3, multiplet=3),
    dict(seed=15, Nrays=5, Nspace=41, Nspect=120, ncol=34, multiplet=4),
    dict(seed=16, Nrays=5, Nspace=41, Nspect=120, ncol=3, multiplet=4),
    dict(seed=17, Nrays=3, Nspace=30, Nspect=150, ncol=33, multiplet=4, phi_compact=True),
]


def _run(lib, prob, block, iters, solver='linear'):
    e = Engine(prob, block.ncol, lib=lib)
    e.set_columns(0, block)
    e.set_formal_solver(solver)
    out = []
    for it in range(iters):
        dJ = e.formal_sol_gamma()
        snap = dict(dJ=dJ, I=e.get(_capi.LSX_I), J=e.get(_capi.LSX_J), G=e.get(_capi.LSX_GAMMA))
        if it >= 2:
            snap['dP'] = e.stat_equil()
            snap['n'] = e.get(_capi.LSX_N)
        out.append(snap)
    e.close()
    return out


def test_toy_problems_are_well_posed_on_the_oracle(oracle_lib):
    prob, block = toy_problem(seed=1)
    o = _run(oracle_lib, prob, block, 6)
    assert all(np.isfinite(s['dJ']) for s in o)
    assert o[-1]['dJ'] < o[1]['dJ'] and o[-1]['dP'] < 0.1
    assert (o[-1]['n'] > 0).all() and (o[-1]['J'] > 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize('kw', CASES, ids=lambda k: '-'.join('%s%s' % (a[:3], b) for a, b in k.items()))
def test_toy_parity(hip_lib, oracle_lib, kw):
    prob, block = toy_problem(**kw)
    h = _run(hip_lib, prob, block, 8)
    o = _run(oracle_lib, prob, block, 8)
    # single call (identical inputs on both sides).  The multiplet cases have an interval whose optical depth lies just above
    # the 5e-4 switch of w2, where w1 = (1 - e) - dtau e cancels to dtau^2 / 2 and a 1-ulp difference between the two exp()
    # implementations is an error of up to 1e-9 of that ray's contribution (tolerances): measured 7e-11
    # -- demonstrated, not assumed (tests/envelope.py): every entry inside 1e-11 + 3 x what a one-ulp change of the oracle's own exp() does
    # to it (round 4 asserted 2e-10 for the multiplets from a measurement)
    envelope.first_call_inside(oracle_lib, prob, block, h[0]['I'], h[0]['J'])
    off, diag = gamma_err(h[0]['G'], o[0]['G'], prob)
    assert off < 1e-10 and diag < 1e-11, (off, diag)
    assert abs(h[0]['dJ'] - o[0]['dJ']) <= 1e-11 * abs(o[0]['dJ'])
    # iterated
    for a, b in zip(h, o):
        assert abs(a['dJ'] - b['dJ']) <= 1e-7 * max(abs(b['dJ']), 1e-3)
    # populations: relative to the largest level population of that depth (a 5-depth column can drive small levels
    # through zero, where a per-element relative error means nothing)
    dn = np.abs(h[-1]['n'] - o[-1]['n']) / np.abs(o[-1]['n']).max(axis=1, keepdims=True)
    assert dn.max() < 1e-8
    assert relerr(h[-1]['J'], o[-1]['J']) < 1e-8
    assert abs(h[-1]['dP'] - o[-1]['dP']) <= 1e-6 * max(abs(o[-1]['dP']), 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [CASES[0], CASES[2], CASES[5], CASES[7], CASES[9]], ids=lambda k: '-'.join('%s%s' % (a[:3], b) for a, b in k.items()))
def test_toy_parity_parabolic_rule(hip_lib, oracle_lib, kw):
    """the parabolic rule (N4) on the same topologies: level cells, chained continua, deep columns, per-wavelength
    scattering -- HIP against the oracle's restatement of the same rule (parity with the reference: unpinned)"""
    prob, block = toy_problem(**kw)
    h = _run(hip_lib, prob, block, 6, solver='parabolic')
    o = _run(oracle_lib, prob, block, 6, solver='parabolic')
    assert relerr(h[0]['I'], o[0]['I']) < 1e-11 and relerr(h[0]['J'], o[0]['J']) < 1e-11
    off, diag = gamma_err(h[0]['G'], o[0]['G'], prob)
    assert off < 1e-10 and diag < 1e-11, (off, diag)
    dn = np.abs(h[-1]['n'] - o[-1]['n']) / np.abs(o[-1]['n']).max(axis=1, keepdims=True)
    assert dn.max() < 1e-8 and relerr(h[-1]['J'], o[-1]['J']) < 1e-8


def _classes_on_ray_serial(lib, eng):
    """-> [(slots, lines, linked, relation)] of the classes the context ran on the ray-serial kernel"""
    import ctypes as C
    f = lib.dll.lsx_hip_class_info
    f.restype = C.c_int32
    f.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    out = (C.c_int64 * 8)()
    n = f(eng._h, -1, out)
    rs = []
    for i in range(n):
        f(eng._h, i, out)
        if out[6] and out[3] > 0:
            rs.append((int(out[0]), int(out[1]), int(out[4]), int(out[5])))
    return rs


RS_CASES = [
    dict(seed=11, Nrays=5, Nspace=82, Nspect=140, ncol=33, chain=False),        # 33 columns: six full groups of five and one of three
    dict(seed=13, Nrays=5, Nspace=41, Nspect=120, ncol=34, multiplet=3),        # odd depth count: the two directions meet in one step
    dict(seed=15, Nrays=5, Nspace=41, Nspect=120, ncol=37, multiplet=4),        # four-line multiplets (generic class beside the ray-serial ones)
    dict(seed=21, Nrays=5, Nspace=3, Nspect=60, ncol=36),                       # the shortest column the rule allows
    dict(seed=22, Nrays=5, Nspace=30, Nspect=90, ncol=41, phi_compact=True),    # ray-independent profiles
]


@pytest.mark.gpu
@pytest.mark.parametrize('kw', RS_CASES, ids=lambda k: '-'.join('%s%s' % (a[:3], b) for a, b in k.items()))
def test_toy_parity_on_the_ray_serial_kernel(hip_lib, oracle_lib, kw):
    """the ray-serial sweep (production: contexts of >= 160 columns) forced onto small toy batches: ragged column groups (>= 32 columns: the per-class launch path), odd and
    minimal depth counts, compact profiles, every tile shape the toy atoms produce with at most two per-ray slots; then the same
    batch with every third column frozen (lsx_set_active_columns): frozen columns keep their J, I, Gamma and populations, the
    others do not notice"""
    kw = dict(kw)
    ncol = kw['ncol']
    prob, block = toy_problem(**kw)
    eh, eo = Engine(prob, ncol, lib=hip_lib, sweep_policy='ray-serial'), Engine(prob, ncol, lib=oracle_lib)
    for e in (eh, eo):
        e.set_columns(0, block)
    for it in range(6):
        dh, do = eh.formal_sol_gamma(), eo.formal_sol_gamma()
        if it == 0:
            assert _classes_on_ray_serial(hip_lib, eh), 'no class of this problem ran on the ray-serial kernel'
            envelope.first_call_inside(oracle_lib, prob, block, eh.get(_capi.LSX_I), eh.get(_capi.LSX_J))      # 1e-11 + the one-ulp-exp envelope
            off, diag = gamma_err(eh.get(_capi.LSX_GAMMA), eo.get(_capi.LSX_GAMMA), prob)
            assert off < 1e-10 and diag < 1e-11, (off, diag)
        assert abs(dh - do) <= 1e-7 * max(abs(do), 1e-3)
        if it >= 2:
            eh.stat_equil(); eo.stat_equil()
    n_o = eo.get(_capi.LSX_N)
    dn = np.abs(eh.get(_capi.LSX_N) - n_o) / np.abs(n_o).max(axis=1, keepdims=True)
    assert dn.max() < 1e-8
    # ---- every third column frozen
    active = np.ones(ncol, dtype=bool)
    active[::3] = False
    before = {w: eh.get(w) for w in (_capi.LSX_J, _capi.LSX_N, _capi.LSX_GAMMA, _capi.LSX_I)}
    for e in (eh, eo):
        e.set_active_columns(active)
        e.formal_sol_gamma(); e.stat_equil()
    for w, v in before.items():
        assert np.array_equal(eh.get(w)[~active], v[~active])
    n_o = eo.get(_capi.LSX_N)
    dn = np.abs(eh.get(_capi.LSX_N) - n_o) / np.abs(n_o).max(axis=1, keepdims=True)
    # (J against the largest J of its column: the three-depth columns have mean intensities that pass through zero)
    Jh, Jo = eh.get(_capi.LSX_J)[active], eo.get(_capi.LSX_J)[active]
    dJ = np.abs(Jh - Jo).reshape(len(Jo), -1).max(axis=1) / np.abs(Jo).reshape(len(Jo), -1).max(axis=1)
    assert dn.max() < 1e-8 and dJ.max() < 1e-7         # (after seven iterations; the single call above agrees to 1e-11)
    eh.close(); eo.close()


@pytest.mark.gpu
@pytest.mark.parametrize('kw', RS_CASES[:3] + RS_CASES[4:], ids=lambda k: '-'.join('%s%s' % (a[:3], b) for a, b in k.items()))
def test_toy_parity_parabolic_rule_compile_time_classes(hip_lib, oracle_lib, kw):
    """five rays and >= 32 columns: the parabolic rule's compile-time tile classes (sweep_tile_par) on the toy topologies -- odd
    depth counts, multiplets (whose four-slot tiles take the generic parabolic instance on their own tile list), compact profiles"""
    prob, block = toy_problem(**kw)
    h = _run(hip_lib, prob, block, 6, solver='parabolic')
    o = _run(oracle_lib, prob, block, 6, solver='parabolic')
    envelope.first_call_inside(oracle_lib, prob, block, h[0]['I'], h[0]['J'], solver='parabolic')
    off, diag = gamma_err(h[0]['G'], o[0]['G'], prob)
    assert off < 1e-10 and diag < 1e-11, (off, diag)
    dn = np.abs(h[-1]['n'] - o[-1]['n']) / np.abs(o[-1]['n']).max(axis=1, keepdims=True)
    assert dn.max() < 1e-8 and relerr(h[-1]['J'], o[-1]['J']) < 1e-8
