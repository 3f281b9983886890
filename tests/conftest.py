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


# ---- the two-rank rehearsals of bench.py (tests/test_bench_launcher.py) are STARTED here, when collection has finished and before
# any test has touched the GPU: a process that has initialised HIP must not start other programs on the GPU box, and by the time
# that test runs this process has.  The test only collects their output.
REHEARSALS = {
    'c3-columns': ['--columns', '200', '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--no-single-column'],
    'c5-response-function': ['--workload', 'c5'],
}
_rehearsal_procs = {}


def pytest_collection_finish(session):
    import shlex
    import subprocess
    import tempfile
    wanted = [it.callspec.params['which'] for it in session.items if it.name.startswith('test_two_rank_rehearsal_on_one_gpu')]
    if not wanted or _rehearsal_procs or session.config.option.collectonly:
        return
    env = dict(os.environ, LSX_BENCH_REHEARSE='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    # ONE child that runs the rehearsals one after the other (the GPU box allows few processes on the card at once: two ranks of one
    # rehearsal beside this process, never four)
    tmp = tempfile.mkdtemp(prefix='lsx_rehearsal_')
    parts = []
    for key in wanted:
        base = os.path.join(tmp, key)
        cmd = ' '.join(shlex.quote(x) for x in [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + REHEARSALS[key])
        parts.append('%s > %s.out 2> %s.err; echo $? > %s.rc' % (cmd, shlex.quote(base), shlex.quote(base), shlex.quote(base)))
    _rehearsal_procs['proc'] = subprocess.Popen(['/bin/sh', '-c', '; '.join(parts)], cwd=ROOT, env=env)
    _rehearsal_procs['dir'] = tmp


def rehearsal_output(key, timeout=900):
    """-> (return code, stdout, stderr) of the rehearsal started at the end of collection"""
    import time
    base = os.path.join(_rehearsal_procs['dir'], key)
    t0 = time.time()
    while not os.path.exists(base + '.rc'):
        if _rehearsal_procs['proc'].poll() is not None and not os.path.exists(base + '.rc'):
            raise RuntimeError('the rehearsal process ended without running %s' % key)
        if time.time() - t0 > timeout:
            raise TimeoutError(key)
        time.sleep(0.5)
    time.sleep(0.1)
    return int(open(base + '.rc').read().strip() or 1), open(base + '.out').read(), open(base + '.err').read()


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
