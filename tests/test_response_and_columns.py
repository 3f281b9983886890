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
