"""D2 (response_fn.py) and per-column convergence: CPU run on the oracle; the same test body runs
on the GPU with -m gpu."""
import numpy as np
import pytest

from conftest import golden, relerr
from lightspinner_amd import fixtures, synth, Engine, _capi, drivers, response
from lightspinner_amd.problem import ColumnBlock


def _rf_case(lib):
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'))
    rf = dict(np.load(golden('rf_ca.npz')))
    ks = [int(k) for k in rf['ks']]
    cols, tags = [base.slice(0, 1)], ['base']
    for k in ks:
        for tag in ('p', 'm'):
            pre = 'k%d%s_' % (k, tag)
            delta = {key[len(pre):]: v for key, v in rf.items() if key.startswith(pre) and
                     key[len(pre):] not in ('I', 'n', 'niter', 'traj_dJ', 'traj_dPops')}
            cols.append(response.apply_delta(prob, base, delta, k, start_n=rf['base_n']))
            tags.append(pre)
    batch = ColumnBlock.concatenate(cols)
    eng = Engine(prob, batch.ncol, lib=lib)
    eng.set_columns(0, batch)
    n_iter = drivers.iterate_mali_columns(eng)
    I, n = eng.get(_capi.LSX_I), eng.get(_capi.LSX_N)
    # every column stops exactly where the reference's own loop stopped for it
    assert n_iter[0] == int(rf['base_niter']) == 46
    assert relerr(I[0], rf['base_I']) < 1e-6 and relerr(n[0], rf['base_n']) < 1e-6
    for c, pre in enumerate(tags[1:], start=1):
        assert n_iter[c] == int(rf[pre + 'niter']), (pre, n_iter[c])
        assert relerr(I[c], rf[pre + 'I']) < 1e-6
        assert relerr(n[c], rf[pre + 'n']) < 1e-6
    # rf[la, k] = (I+ - I-) / I_base at mu index -1 (response_fn.py:67)
    for q, k in enumerate(ks):
        mine = (I[1 + 2 * q][:, -1] - I[2 + 2 * q][:, -1]) / I[0][:, -1]
        ref = (rf['k%dp_I' % k][:, -1] - rf['k%dm_I' % k][:, -1]) / rf['base_I'][:, -1]
        assert np.allclose(mine, ref, rtol=0, atol=2e-6 * np.max(np.abs(ref)))
    eng.close()


def test_response_function_columns_oracle(oracle_lib):
    _rf_case(oracle_lib)


@pytest.mark.gpu
def test_response_function_columns_gpu(hip_lib):
    _rf_case(hip_lib)


def _freeze_case(lib):
    """frozen columns are neither read nor written, and report dJ = dPops = 0"""
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'))
    eng = Engine(prob, 3, lib=lib)
    eng.set_columns(0, ColumnBlock.concatenate([base, base, base]))
    eng.formal_sol_gamma()
    J1 = eng.get(_capi.LSX_J)
    eng.set_active_columns([1, 0, 1])
    dJ = eng.formal_sol_gamma()
    J2 = eng.get(_capi.LSX_J)
    assert np.array_equal(J2[1], J1[1]) and not np.array_equal(J2[0], J1[0]) and np.array_equal(J2[0], J2[2])
    d = eng.get(_capi.LSX_DJ_COL)
    assert d[1] == 0.0 and d[0] == d[2] == dJ
    eng.set_active_columns(None)
    eng.formal_sol_gamma()
    assert not np.array_equal(eng.get(_capi.LSX_J)[1], J1[1])
    with pytest.raises(ValueError):
        eng.set_active_columns([1, 0])
    eng.close()


def test_freeze_columns_oracle(oracle_lib):
    _freeze_case(oracle_lib)


@pytest.mark.gpu
def test_freeze_columns_gpu(hip_lib):
    _freeze_case(hip_lib)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['ray-per-lane', 'ray-serial'])
def test_freeze_columns_of_a_many_level_atom_on_the_per_class_path_gpu(hip_lib, oracle_lib, mode):
    """the column mask through the kernels a many-level atom takes at many columns (round 5: the big-set instances of the
    column-mapped fast-continuum epilogue, the thread-per-column Gamma epilogue): MgII, 33 perturbed columns, every third one
    frozen after two calls -- frozen columns keep J, Gamma and populations bit for bit and report dJ = dPops = 0, the others
    go on like the oracle's"""
    prob, base, raw = fixtures.load_problem_npz(golden('falc_mg.npz'), phi_compact=False)
    ncol = 33
    blk, (aD, vB, vlos) = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=77, vlos_sigma=1.5e3)
    engs = []
    for lib in (hip_lib, oracle_lib):
        e = Engine(prob, ncol, lib=lib, **(dict(sweep_policy=mode) if lib is hip_lib else {}))
        e.set_columns(0, blk)
        e.set_line_profiles(0, aD, vB, vlos)
        engs.append(e)
    hip, ora = engs
    oracle_lib.dll.lsx_oracle_set_threads(ora._h, 8)
    for e in engs:
        e.formal_sol_gamma(); e.formal_sol_gamma()
    keep = {q: hip.get(q).copy() for q in (_capi.LSX_J, _capi.LSX_GAMMA, _capi.LSX_N)}
    mask = np.array([c % 3 != 1 for c in range(ncol)], dtype=np.uint8)
    frozen = np.flatnonzero(mask == 0)
    for e in engs:
        e.set_active_columns(mask)
    for it in range(3):
        dJ, dJo = hip.formal_sol_gamma(), ora.formal_sol_gamma()
        dP, dPo = hip.stat_equil(), ora.stat_equil()
        assert dJ == pytest.approx(dJo, rel=1e-6) and dP == pytest.approx(dPo, rel=1e-6)
    for q, v in keep.items():
        now = hip.get(q)
        assert np.array_equal(now[frozen], v[frozen]) and not np.array_equal(now[0], v[0]), q
    assert np.all(hip.get(_capi.LSX_DJ_COL)[frozen] == 0.0) and np.all(hip.get(_capi.LSX_DPOPS_COL)[frozen] == 0.0)
    live = np.flatnonzero(mask)
    assert relerr(hip.get(_capi.LSX_N)[live], ora.get(_capi.LSX_N)[live]) < 1e-7
    assert relerr(hip.get(_capi.LSX_J)[live], ora.get(_capi.LSX_J)[live], floor=1e-300) < 1e-7
    for e in engs:
        e.close()


def _all_depths_fixture():
    import os
    p = golden('rf_ca_inputs.npz')
    if not os.path.exists(p):
        pytest.skip('rf_ca_inputs.npz not generated')
    return dict(np.load(p))


def test_all_depth_inputs_agree_with_the_three_depth_fixture():
    """the two generators of make_golden.py (gen_rf: 3 depths with reference outputs, gen_rf_inputs: all depths, inputs
    only) describe the same perturbed atmospheres"""
    fx, rf = _all_depths_fixture(), dict(np.load(golden('rf_ca.npz')))
    assert int(fx['Nspace']) == 82
    for k in [int(k) for k in rf['ks']]:
        for tag in ('p', 'm'):
            a, b = response.deltas_of(fx, k, tag), response.deltas_of(rf, k, tag)
            assert set(a) == set(b)
            for key in a:
                assert np.array_equal(a[key], b[key]), (k, tag, key)


def _rf_all_depths(lib):
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'))
    fx, rf = _all_depths_fixture(), dict(np.load(golden('rf_ca.npz')))
    out = response.run_response_function(prob, base, fx, range(82), lib=lib)
    assert out['n_iter_base'] == 46 and out['rf'].shape == (prob.Nspect, 82)
    assert out['n_iter'].min() >= 1 and out['n_iter'].max() < 46        # warm started: a handful of iterations each
    for k in [int(k) for k in rf['ks']]:
        ref = (rf['k%dp_I' % k][:, -1] - rf['k%dm_I' % k][:, -1]) / rf['base_I'][:, -1]
        assert np.allclose(out['rf'][:, k], ref, rtol=0, atol=2e-6 * np.max(np.abs(ref)))
        assert out['n_iter'][2 * k] == int(rf['k%dp_niter' % k]) and out['n_iter'][2 * k + 1] == int(rf['k%dm_niter' % k])
    assert np.all(np.isfinite(out['rf']))


def test_response_function_all_depths_oracle(oracle_lib):
    _rf_all_depths(oracle_lib)


@pytest.mark.gpu
def test_response_function_all_depths_gpu(hip_lib):
    _rf_all_depths(hip_lib)
