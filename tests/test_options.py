"""Explicit options and the options signature (include/lsx.h, round 5): every switch that decides how a context associates its sums
is an argument of lsx_create_with_options (the LSX_* environment variables are defaults that an explicit entry overrides), is reported
by lsx_effective_options and hashed by lsx_options_signature; a sharded job exchanges the signatures once and refuses a rank that was
made differently (parallel.check_same_options).  CPU side: the product's own host-side plan (liblsx_host.so, `make host`) and the
oracle; the GPU side is tests/test_options_gpu.py."""
import os
import shutil
import socket
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden
from lightspinner_amd import fixtures, Engine, _capi

CSRC = os.path.join(ROOT, 'lightspinner_amd', 'csrc')
pytestmark = pytest.mark.skipif(shutil.which('g++') is None, reason='no host compiler')


@pytest.fixture(scope='module')
def host():
    subprocess.check_call(['make', '-s', '-C', CSRC, 'host'])
    from san_driver import HostOnly
    return HostOnly(os.path.join(CSRC, 'liblsx_host.so'))


def _kv(s):
    return dict(x.split('=', 1) for x in s.split(';'))


def test_explicit_options_override_the_environment_and_show_in_the_signature(host, monkeypatch):
    for v in ('LSX_NO_LINKED', 'LSX_RS_MAX_NPT', 'LSX_TILER', 'LSX_NO_RS', 'LSX_CLASS_CHUNK'):
        monkeypatch.delenv(v, raising=False)
    prob = fixtures.load_problem_npz(golden('falc_cah.npz'), phi_compact=False)[0]
    h0, s0 = host.signature(prob)
    kv = _kv(s0)
    assert h0 != 0 and kv['linked'] == '1' and kv['tiler'] == 'dp' and kv['rs'] == '1' and kv['rs_max_npt'] == '2'
    assert '1.1.1.0:34s' in kv['classes']                    # the 34 hydrogen-line tiles with linked continua, on the ray-serial list
    # an explicit entry changes the plan and the hash; the same entry twice gives the same hash
    h1, s1 = host.signature(prob, 'linked=0')
    assert h1 not in (0, h0) and _kv(s1)['linked'] == '0' and '3.1.0.0' in _kv(s1)['classes']
    assert host.signature(prob, 'linked=0')[0] == h1
    # the environment is a DEFAULT ...
    monkeypatch.setenv('LSX_NO_LINKED', '1')
    assert host.signature(prob) == (h1, s1)
    # ... that an explicit entry overrides
    assert host.signature(prob, 'linked=1') == (h0, s0)
    monkeypatch.delenv('LSX_NO_LINKED')
    # launch-shape options show as well (same bits, tested on the GPU; the signature does not try to know that)
    h2, s2 = host.signature(prob, 'class_chunk=9, rs_max_npt=1')
    assert h2 not in (0, h0, h1) and _kv(s2)['class_chunk'] == '9' and _kv(s2)['classes'].count('1.1.1.0:') == 4
    # malformed lists are refused with a message, not ignored
    for bad in ('linked', 'linked=', 'linked=2', 'nonsense=1', 'rs_max_npt=7', 'tiler=zigzag'):
        h, msg = host.signature(prob, bad)
        assert h == 0 and 'options' in msg, (bad, msg)


def test_oracle_accepts_and_ignores_well_formed_lists(oracle_lib):
    prob, block, _ = fixtures.load_problem_npz(golden('falc_ca.npz'))
    e = Engine(prob, 1, lib=oracle_lib, options='linked=0,rs_max_npt=1')
    assert e.effective_options() == 'backend=oracle-c'
    f = Engine(prob, 1, lib=oracle_lib, options=dict(tiler='natural'))
    assert e.options_signature() == f.options_signature() != 0
    with pytest.raises(_capi.LsxError):
        Engine(prob, 1, lib=oracle_lib, options='linked')
    with pytest.raises(_capi.LsxError):
        small = (e.lib.dll.lsx_effective_options, )
        import ctypes as C
        buf = C.create_string_buffer(4)
        e.lib.check(e.lib.dll.lsx_effective_options(e._h, buf, 4))
    e.close(); f.close()


# ---- a rank made differently is refused --------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _PlanEngine:
    """what parallel.check_same_options needs of an engine, from the product's host-side plan (no GPU in this container)"""

    def __init__(self, host, prob, options=None):
        self.sig, self.text = host.signature(prob, options)
        assert self.sig != 0, self.text

    def options_signature(self):
        return self.sig

    def effective_options(self):
        return self.text


def _worker(rank, world, port, odd_rank, how, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for v in ('LSX_NO_LINKED', 'LSX_RS_MAX_NPT'):
        os.environ.pop(v, None)
    explicit = None
    if rank == odd_rank:
        if how == 'env':
            os.environ['LSX_NO_LINKED'] = '1'           # the round-4 failure mode: one rank's environment differs
        elif how == 'list':
            explicit = 'rs_max_npt=1'
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from san_driver import HostOnly
    from lightspinner_amd.parallel import check_same_options, OptionsMismatch, MaxReducer
    host = HostOnly(os.path.join(CSRC, 'liblsx_host.so'))
    prob = fixtures.load_problem_npz(golden('falc_cah.npz'), phi_compact=False)[0]
    eng = _PlanEngine(host, prob, explicit)
    try:
        check_same_options(eng)
        verdict = 'accepted'
    except OptionsMismatch as e:
        verdict = 'refused: %s' % e
    with open(os.path.join(out_dir, 'rank%d.txt' % rank), 'w') as f:
        f.write(verdict)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('how,odd_rank', [('none', -1), ('env', 1), ('list', 0)])
def test_a_rank_with_a_different_environment_is_refused_on_every_rank(host, tmp_path, how, odd_rank):
    import torch.multiprocessing as mp
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), odd_rank, how, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    out = [open(os.path.join(str(tmp_path), 'rank%d.txt' % r)).read() for r in range(world)]
    if how == 'none':
        assert out == ['accepted', 'accepted']
    else:
        for r, text in enumerate(out):
            assert text.startswith('refused') and 'ranks [1] differ from rank 0' in text and 'rank %d:' % r in text, text
        # each rank names its own effective options: the odd one shows what it was made with
        assert ('linked=0' if how == 'env' else 'rs_max_npt=1') in out[odd_rank]
