"""The MALI loop without a host round trip per iteration (include/lsx.h: lsx_sync_begin / lsx_sync_end /
lsx_formal_sol_gamma_speculative / lsx_discard_formal_sol; drivers.iterate_mali_engine, drivers.mali_steps) against the plain
loop of test.py:20-29: the same iterations, the same monitors, and every result -- J, I, Gamma, populations -- the same bits.
CPU: on the oracle (the ABI is shared); GPU: on the HIP library, single column (the fused launch) and a batch (per-class path)."""
import numpy as np
import pytest

from conftest import golden
from lightspinner_amd import fixtures, synth, Engine, _capi, drivers

RESULTS = (_capi.LSX_J, _capi.LSX_I, _capi.LSX_GAMMA, _capi.LSX_N, _capi.LSX_DJ_COL, _capi.LSX_DPOPS_COL)


def _columns(ncol, seed=5):
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'))
    if ncol == 1:
        return prob, base
    blk, _ = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=seed, vlos_sigma=0.0)
    return prob, blk


def _both_loops(lib, ncol, **kw):
    prob, blk = _columns(ncol)
    out = []
    for pipelined in (False, True):
        e = Engine(prob, ncol, lib=lib)
        e.set_columns(0, blk)
        h = drivers.iterate_mali_engine(e, pipelined=pipelined, **kw)
        out.append((h, [e.get(w) for w in RESULTS]))
        e.close()
    return out


def _same(out):
    (h0, r0), (h1, r1) = out
    assert h0.n_iter == h1.n_iter and h0.converged == h1.converged
    assert h0.dJ == h1.dJ
    assert np.array_equal(np.array(h0.dPops), np.array(h1.dPops), equal_nan=True)
    for a, b in zip(r0, r1):
        assert np.array_equal(a, b)
    return h0


def test_pipelined_loop_on_the_oracle_is_the_plain_loop():
    import oracle
    lib = oracle.load()
    h = _same(_both_loops(lib, 1))
    assert h.converged and h.n_iter == 46                      # the reference's own count (tests/golden/falc_ca.npz)
    h = _same(_both_loops(lib, 3, max_iter=7))                  # stopped by max_iter: nothing speculative is left behind
    assert h.n_iter == 7 and not h.converged


def test_discard_and_its_preconditions_on_the_oracle():
    import oracle
    _discard_semantics(oracle.load())


def _discard_semantics(lib):
    prob, blk = _columns(2)
    e = Engine(prob, 2, lib=lib)
    e.set_columns(0, blk)
    with pytest.raises(_capi.LsxError, match='not a speculative'):
        e.discard_formal_sol()
    for _ in range(4):
        e.formal_sol_gamma()
    e.stat_equil()
    e.formal_sol_gamma()
    with pytest.raises(_capi.LsxError, match='not a speculative'):
        e.discard_formal_sol()                                  # a plain call cannot be taken back
    before = [e.get(w) for w in RESULTS]
    e.formal_sol_gamma_speculative()
    e.sync()
    assert not np.array_equal(e.get(_capi.LSX_J), before[0])     # the speculative call is what lsx_get shows ...
    e.discard_formal_sol()
    for w, b in zip(RESULTS, before):
        assert np.array_equal(e.get(w), b)                      # ... until it is taken back
    with pytest.raises(_capi.LsxError, match='not a speculative'):
        e.discard_formal_sol()                                  # once
    # a speculative call that is built upon is an ordinary one: same populations as the plain sequence
    e2 = Engine(prob, 2, lib=lib)
    e2.set_columns(0, blk)
    for _ in range(4):
        e2.formal_sol_gamma()
    e2.stat_equil()
    e2.formal_sol_gamma()
    e.formal_sol_gamma_speculative(); e.stat_equil()
    e2.formal_sol_gamma(); e2.stat_equil()
    with pytest.raises(_capi.LsxError, match='not a speculative'):
        e.discard_formal_sol()
    for w in RESULTS:
        assert np.array_equal(e.get(w), e2.get(w))
    # one read-back in flight at a time; lsx_sync_end without one is lsx_sync
    e.sync_begin()
    with pytest.raises(_capi.LsxError, match='not been collected'):
        e.sync_begin()
    a = e.sync_end()
    assert a == e.sync_end() == e.sync()
    # frozen columns: no speculation (the loop then simply does not look ahead)
    e.set_active_columns(np.array([True, False]))
    with pytest.raises(_capi.LsxError, match='frozen'):
        e.formal_sol_gamma_speculative()
    h = drivers.iterate_mali_engine(e, max_iter=3)
    assert h.n_iter == 3
    e.close(); e2.close()


def test_mali_steps_are_mali_step_on_the_oracle():
    import oracle
    _steps(oracle.load(), 2)


def _steps(lib, ncol):
    prob, blk = _columns(ncol)
    e1, e2 = Engine(prob, ncol, lib=lib), Engine(prob, ncol, lib=lib)
    for e in (e1, e2):
        e.set_columns(0, blk)
    a = list(drivers.mali_steps(e1, 6, n_lambda_only=3, lookahead=True))
    b = [drivers.mali_step(e2, it > 3) for it in range(1, 7)]
    assert a == b
    for w in RESULTS:
        assert np.array_equal(e1.get(w), e2.get(w))
    e1.close(); e2.close()


@pytest.mark.gpu
@pytest.mark.parametrize('ncol', [1, 40])
def test_pipelined_loop_on_hip_is_the_plain_loop(hip_lib, ncol):
    h = _same(_both_loops(hip_lib, ncol, max_iter=60))
    assert h.converged and (ncol > 1 or h.n_iter == 46)


@pytest.mark.gpu
def test_discard_and_its_preconditions_on_hip(hip_lib):
    _discard_semantics(hip_lib)


@pytest.mark.gpu
def test_mali_steps_are_mali_step_on_hip(hip_lib):
    _steps(hip_lib, 1)
    _steps(hip_lib, 33)
