"""Enqueue-only sequences of the C ABI (include/lsx.h: the asynchronous forms, lsx_sync_begin / lsx_sync_end, the speculative
formal solution) must not lose monitors between calls.  rh_method.py:741-745 returns a statistical equilibrium's maximum change
from the call itself; over the asynchronous ABI the caller may have enqueued the next formal solution before it reads that
maximum (and the singular flag behind scipy's LinAlgError, rh_method.py:739) -- the library has to keep them until they are read.

CPU: the oracle (the ABI is shared, its calls are synchronous); GPU: the HIP library against the same expectations."""
import numpy as np
import pytest

from conftest import golden
from lightspinner_amd import fixtures, synth, Engine, _capi


def _engines(lib, ncol=3, n=2):
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'))
    blk, _ = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=9, vlos_sigma=0.0)
    out = []
    for _ in range(n):
        e = Engine(prob, ncol, lib=lib)
        e.set_columns(0, blk)
        out.append(e)
    return out


def _fs_se_fs_sync(lib):
    """`FS; SE; FS; lsx_sync` reports the formal solution's dJ AND the statistical equilibrium's dPops (per column too)"""
    a, b = _engines(lib)
    for e in (a, b):
        for _ in range(4):
            e.formal_sol_gamma()
    # plain sequence
    dP_plain = b.stat_equil()
    dPcol_plain = b.get(_capi.LSX_DPOPS_COL)
    dJ_plain = b.formal_sol_gamma()
    # everything enqueued, one read at the end
    a.stat_equil_async()
    a.formal_sol_gamma_async()
    dJ, dP = a.sync()
    assert dP_plain > 0.0
    assert (dJ, dP) == (dJ_plain, dP_plain)
    assert np.array_equal(a.get(_capi.LSX_DPOPS_COL), dPcol_plain)
    for w in (_capi.LSX_J, _capi.LSX_I, _capi.LSX_GAMMA, _capi.LSX_N):
        assert np.array_equal(a.get(w), b.get(w))
    # the statistical equilibrium that follows starts from cleared maxima: same value as in the plain sequence
    assert a.stat_equil() == b.stat_equil()
    # ... and the blocking formal solution right behind an enqueued stat_equil keeps its dPops for the next lsx_sync
    a.formal_sol_gamma(); b.formal_sol_gamma()
    dP_plain = b.stat_equil()
    dJ_plain = b.formal_sol_gamma()
    a.stat_equil_async()
    assert a.formal_sol_gamma() == dJ_plain
    assert a.sync() == (dJ_plain, dP_plain)
    a.close(); b.close()


def _singular_is_not_lost(lib):
    """a singular system (Gamma still zero, rh_method.py:739) is reported even if a formal solution was enqueued behind it"""
    (e,) = _engines(lib, n=1)
    with pytest.raises(np.linalg.LinAlgError):
        e.stat_equil_async()            # (the oracle computes here, the HIP library at the read-back)
        e.formal_sol_gamma_async()
        e.sync()
    e.close()


def _discard_restores_the_pending_call(lib):
    """`FS_async; FS_speculative; discard; lsx_sync` reports the FIRST call's dJ; a read-back begun after the speculative call has
    to be collected before that call can be discarded"""
    a, b = _engines(lib)
    for e in (a, b):
        for _ in range(3):
            e.formal_sol_gamma()
    a.formal_sol_gamma_async()
    a.formal_sol_gamma_speculative()
    a.discard_formal_sol()
    dJ_a, _ = a.sync()
    assert dJ_a == b.formal_sol_gamma()
    for w in (_capi.LSX_J, _capi.LSX_I, _capi.LSX_GAMMA, _capi.LSX_DJ_COL):
        assert np.array_equal(a.get(w), b.get(w))
    # the speculative call's monitors were read with lsx_sync, then it is taken back: lsx_sync is the previous call's again
    dJ_prev = dJ_a
    a.formal_sol_gamma_speculative()
    dJ_spec, _ = a.sync()
    assert dJ_spec != dJ_prev
    a.discard_formal_sol()
    assert a.sync()[0] == dJ_prev
    # a read-back begun AFTER the speculative call: collect it first
    a.formal_sol_gamma_speculative()
    a.sync_begin()
    with pytest.raises(_capi.LsxError, match='in flight'):
        a.discard_formal_sol()
    assert a.sync_end()[0] == dJ_spec
    a.discard_formal_sol()
    assert a.sync()[0] == dJ_prev
    for w in (_capi.LSX_J, _capi.LSX_I, _capi.LSX_GAMMA, _capi.LSX_DJ_COL):
        assert np.array_equal(a.get(w), b.get(w))
    a.close(); b.close()


CASES = [_fs_se_fs_sync, _singular_is_not_lost, _discard_restores_the_pending_call]


@pytest.mark.parametrize('case', CASES, ids=[c.__name__.strip('_') for c in CASES])
def test_on_the_oracle(oracle_lib, case):
    case(oracle_lib)


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES, ids=[c.__name__.strip('_') for c in CASES])
def test_on_hip(hip_lib, case):
    case(hip_lib)


@pytest.mark.gpu
def test_on_hip_many_columns(hip_lib):
    """the many-column Gamma epilogue (k_gamma_finish) takes the same care as the small-batch one"""
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'))
    blk, _ = synth.perturbed_columns(prob, base, raw, ncol=40, seed=9, vlos_sigma=0.0)
    a, b = Engine(prob, 40, lib=hip_lib), Engine(prob, 40, lib=hip_lib)
    for e in (a, b):
        e.set_columns(0, blk)
        for _ in range(4):
            e.formal_sol_gamma()
    dP = b.stat_equil()
    dJ = b.formal_sol_gamma()
    a.stat_equil_async(); a.formal_sol_gamma_async()
    assert a.sync() == (dJ, dP) and dP > 0.0
    assert np.array_equal(a.get(_capi.LSX_N), b.get(_capi.LSX_N))
    a.close(); b.close()
