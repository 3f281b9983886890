"""D2 (response_fn.py) and per-column convergence: CPU run on the oracle; the same test body runs
on the GPU with -m gpu."""
import numpy as np
import pytest

from conftest import golden, relerr
from lightspinner_amd import fixtures, Engine, _capi, drivers, response
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
