import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle_lib():
    """The CPU restatement (checker).  tests/ is one of the three places allowed to load it."""
    import oracle
    return oracle.load()


@pytest.fixture(scope='session')
def hip_lib():
    """The product library; GPU tests fail loudly (not skip) if it is not built."""
    from lightspinner_amd import _capi
    return _capi.load_hip_library()


def golden(name):
    return os.path.join(GOLDEN, name)


def relerr(a, b, floor=1e-300):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def gamma_err(G, Gref, prob):
    """SURVEY 8d bar for Gamma: off-diagonal relative; diagonal relative to the
    largest entry of its column (the diagonal is a difference of large numbers)."""
    off, diag = 0.0, 0.0
    for a in range(prob.Natoms):
        nl = prob.Nlevel[a]
        o = prob.lev2_off[a]
        g = G[..., o:o + nl * nl, :].reshape(G.shape[:-2] + (nl, nl, prob.Nspace))
        r = Gref[..., o:o + nl * nl, :].reshape(g.shape)
        colmax = np.max(np.abs(r), axis=-3, keepdims=True)  # max over row index l of Gamma[l, i, k]
        for i in range(nl):
            for j in range(nl):
                if i == j:
                    diag = max(diag, float(np.max(np.abs(g[..., i, j, :] - r[..., i, j, :]) / colmax[..., 0, j, :])))
                else:
                    den = np.maximum(np.abs(r[..., i, j, :]), 1e-300)
                    m = r[..., i, j, :] != 0
                    if np.any(m):
                        off = max(off, float(np.max((np.abs(g[..., i, j, :] - r[..., i, j, :]) / den)[m])))
                    if np.any(~m):
                        assert np.all(g[..., i, j, :][~m] == 0)
    return off, diag
