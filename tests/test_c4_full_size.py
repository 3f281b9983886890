"""BASELINE configuration C4 at its FULL size in one context: 10 000 Ca+H columns x 82 depths x 777
wavelengths on ONE GPU (70 GB of inputs; the configuration is defined as 1 250 columns per GPU over eight, which is what the other tests
and `bench.py`'s c4_share exercise -- this is the same problem unsharded).  The oracle would need an hour for it, so the check is the
size-independent property of tests/test_production_classes.py: columns are independent 1-D problems (response_fn.py:61-65),
hence a batch made of copies of ten distinct columns in a scrambled order must give every copy the bits its original gets in a batch
of 37 -- whatever its position among 10 000, its neighbours or its place inside a five-column wavefront -- and the 37-column batch is
checked against the oracle."""
import numpy as np
import pytest

from conftest import golden, relerr
from lightspinner_amd import fixtures, synth, Engine, _capi

pytestmark = pytest.mark.gpu

NCOL, NUNIQ, NSMALL, CHUNK = 10000, 10, 37, 500


def test_c4_ten_thousand_columns_in_one_context(hip_lib, oracle_lib):
    import torch
    free, total = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip('needs ~100 GB of device memory, %.0f GB free' % (free / 1e9))
    prob, base, raw = fixtures.load_problem_npz(golden('falc_cah.npz'), phi_compact=False)
    blk, prof = synth.perturbed_columns(prob, base, raw, ncol=NUNIQ, seed=77, vlos_sigma=2.0e3)
    rng = np.random.default_rng(5)
    src = np.concatenate([np.arange(NUNIQ), rng.integers(0, NUNIQ, NCOL - NUNIQ)])
    pick = lambda idx: (type(blk).concatenate([blk.slice(int(q), int(q) + 1) for q in idx]), tuple(p[idx] for p in prof))
    small = Engine(prob, NSMALL, lib=hip_lib, policy_columns=NCOL)      # the kernels a context of 10 000 columns runs
    big = Engine(prob, NCOL, lib=hip_lib)
    synth.load_columns(small, *pick(src[:NSMALL]))
    for c0 in range(0, NCOL, CHUNK):                                     # (the host never holds more than 500 columns of inputs)
        b, p = pick(src[c0:c0 + CHUNK])
        synth.load_columns(big, b, p, col0=c0)
    assert big.sweep_policy() == small.sweep_policy() == 'ray-serial'
    ora = Engine(prob, NUNIQ, lib=oracle_lib)
    synth.load_columns(ora, blk, prof)
    oracle_lib.dll.lsx_oracle_set_threads(ora._h, 10)
    for it in range(1, 6):                                               # test.py:20-29: three formal solutions, then two full iterations
        dJs, dJb = small.formal_sol_gamma(), big.formal_sol_gamma()
        ora.formal_sol_gamma()
        assert dJs == dJb
        if it > 3:
            assert small.stat_equil() == big.stat_equil()
            ora.stat_equil()
    first = np.array([int(np.nonzero(src[:NSMALL] == q)[0][0]) for q in range(NUNIQ)])
    for what in (_capi.LSX_I, _capi.LSX_N, _capi.LSX_DJ_COL, _capi.LSX_DPOPS_COL):
        a = small.get(what)
        for c0 in range(0, NCOL, 2000):                                  # (read back in pieces)
            b = big.get(what, c0, min(2000, NCOL - c0))
            assert np.array_equal(b, a[first][src[c0:c0 + 2000]]), (what, c0)
    for c0 in (0, 4321, NCOL - 700):                                     # J and Gamma of three windows: first, middle, last columns
        for what in (_capi.LSX_J, _capi.LSX_GAMMA):
            assert np.array_equal(big.get(what, c0, 700), small.get(what)[first][src[c0:c0 + 700]]), (what, c0)
    assert relerr(small.get(_capi.LSX_N)[first], ora.get(_capi.LSX_N)) < 1e-8 and relerr(small.get(_capi.LSX_I)[first], ora.get(_capi.LSX_I)) < 1e-8
    small.close(); big.close(); ora.close()
