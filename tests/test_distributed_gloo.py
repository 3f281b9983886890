"""The N > 1 path on CPU: 2 processes, gloo backend, columns block-partitioned over ranks, the only
exchange being the all-reduce(MAX) of (dJ, dPops).  The oracle stands in for the GPU
library (same ABI) so the sharding / reduction logic is what is tested here."""
import os
import socket

import numpy as np
import pytest

from conftest import golden, ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ncol_total, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle
    from lightspinner_amd import fixtures, synth, Engine, _capi, drivers
    from lightspinner_amd.parallel import shard_columns, MaxReducer
    lib = oracle.load()
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'), phi_compact=False)
    first, n = shard_columns(ncol_total, rank, world)
    batch = synth.perturbed_columns(prob, base, raw, ncol=n, seed=99, first=first)
    eng = Engine(prob, n, lib=lib)
    eng.set_columns(0, batch)

    class A:
        def formal_sol_gamma_matrices(self): return eng.formal_sol_gamma()
        def stat_equil(self): return eng.stat_equil()
    h = drivers.iterate_mali(A(), max_iter=8, reduce_max=MaxReducer())
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), first=first, n=eng.get(_capi.LSX_N), J=eng.get(_capi.LSX_J),
             I=eng.get(_capi.LSX_I), dJ=np.array(h.dJ), dP=np.array(h.dPops))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    ncol_total, world = 5, 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, ncol_total, str(tmp_path)), nprocs=world, join=True, start_method='spawn')

    from lightspinner_amd import fixtures, synth, Engine, _capi, drivers
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'), phi_compact=False)
    batch = synth.perturbed_columns(prob, base, raw, ncol=ncol_total, seed=99)
    eng = Engine(prob, ncol_total, lib=oracle_lib)
    eng.set_columns(0, batch)

    class A:
        def formal_sol_gamma_matrices(self): return eng.formal_sol_gamma()
        def stat_equil(self): return eng.stat_equil()
    h = drivers.iterate_mali(A(), max_iter=8)
    n, J, I = eng.get(_capi.LSX_N), eng.get(_capi.LSX_J), eng.get(_capi.LSX_I)
    seen = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        f, cnt = int(d['first']), d['n'].shape[0]
        # columns are independent: per-column results are bit-identical however they are sharded
        assert np.array_equal(d['n'], n[f:f + cnt]) and np.array_equal(d['J'], J[f:f + cnt]) and np.array_equal(d['I'], I[f:f + cnt])
        # the reduced convergence monitors equal the single-process maxima on every rank
        assert np.array_equal(d['dJ'], np.array(h.dJ)) and np.array_equal(d['dP'][3:], np.array(h.dPops)[3:])
        seen += cnt
    assert seen == ncol_total


@pytest.mark.gpu
def test_max_reducer_over_rccl_single_rank():
    """the RCCL path of the convergence reduction (device tensor, ReduceOp.MAX, NaN flag) on a one-rank communicator --
    what the multi-GPU bench uses per iteration; more ranks need more GPUs than the test box has"""
    import socket
    import torch
    import torch.distributed as dist
    from lightspinner_amd.parallel import MaxReducer
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        r = MaxReducer(device=torch.device('cuda', 0))
        assert not r.active                      # one rank: the reduction is the identity and is skipped
        r.active = True                          # force the collective path
        assert r(0.25, 1.5e-3) == (0.25, 1.5e-3)
        a, b = r(float('nan'), 1.0)
        assert np.isnan(a) and np.isnan(b)       # NaN wins, like ndarray.max (rh_method.py:706)
        assert r(3.0, 0.0) == (3.0, 0.0)
    finally:
        dist.destroy_process_group()
