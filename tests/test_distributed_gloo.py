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
    eng = Engine(prob, n, lib=lib)
    synth.load_columns(eng, *synth.perturbed_columns(prob, base, raw, ncol=n, seed=99, first=first))

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
    eng = Engine(prob, ncol_total, lib=oracle_lib)
    synth.load_columns(eng, *synth.perturbed_columns(prob, base, raw, ncol=ncol_total, seed=99))

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


def _worker_engine_path(rank, world, port, ncol_total, out_dir):
    """the same sharded loop through drivers.iterate_mali_engine + MaxReducer.engine (lsx_monitors -> all-reduce ->
    one read-back), which is what bench.py runs per step on several GPUs"""
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
    eng = Engine(prob, n, lib=lib)
    synth.load_columns(eng, *synth.perturbed_columns(prob, base, raw, ncol=n, seed=99, first=first))
    h = drivers.iterate_mali_engine(eng, reducer=MaxReducer(), max_iter=7, pipelined=True)     # the look-ahead loop over the ranks (engine_begin / engine_end) against the plain single-process loop below
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), first=first, n=eng.get(_capi.LSX_N), J=eng.get(_capi.LSX_J),
             dJ=np.array(h.dJ), dP=np.array(h.dPops))
    dist.barrier()
    dist.destroy_process_group()


def test_four_rank_uneven_shards_engine_monitor_path(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    ncol_total, world = 6, 4            # shards of 2, 2, 1, 1 columns
    port = _free_port()
    mp.start_processes(_worker_engine_path, args=(world, port, ncol_total, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    from lightspinner_amd import fixtures, synth, Engine, _capi, drivers
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'), phi_compact=False)
    eng = Engine(prob, ncol_total, lib=oracle_lib)
    synth.load_columns(eng, *synth.perturbed_columns(prob, base, raw, ncol=ncol_total, seed=99))
    h = drivers.iterate_mali_engine(eng, max_iter=7)
    n, J = eng.get(_capi.LSX_N), eng.get(_capi.LSX_J)
    sizes = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        f, cnt = int(d['first']), d['n'].shape[0]
        sizes.append(cnt)
        assert np.array_equal(d['n'], n[f:f + cnt]) and np.array_equal(d['J'], J[f:f + cnt])
        assert np.array_equal(d['dJ'], np.array(h.dJ)) and np.array_equal(d['dP'][3:], np.array(h.dPops)[3:])
    assert sizes == [2, 2, 1, 1]


def _worker_rf(rank, world, port, out_dir):
    """C5 over ranks: response.run_response_function with the perturbed columns block partitioned, per-column
    convergence, all_done = AND over ranks, results gathered on every rank"""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle
    from lightspinner_amd import fixtures, response
    from lightspinner_amd.parallel import AllDone
    lib = oracle.load()
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'))
    rf = dict(np.load(golden('rf_ca.npz')))
    ks = [int(k) for k in rf['ks']]
    out = response.run_response_function(prob, base, rf, ks, lib=lib, rank=rank, world=world, all_done=AllDone(),
                                         gather=response.gather_over_ranks())
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), rf=out['rf'], n_iter=out['n_iter'], I=out['I'], shard=np.array(out['shard']))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [4, 8])
def test_response_function_sharded_over_ranks(tmp_path, oracle_lib, world):
    """6 perturbed columns (3 depths x +-) over 4 ranks (2, 2, 1, 1) and over 8 ranks (six with one column, two with
    none): every rank ends with the full response function, bit-identical to the single-process run, and every column
    took exactly the iterations the reference's own loop took for it"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.start_processes(_worker_rf, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    from lightspinner_amd import fixtures, response
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'))
    rf = dict(np.load(golden('rf_ca.npz')))
    ks = [int(k) for k in rf['ks']]
    one = response.run_response_function(prob, base, rf, ks, lib=oracle_lib)
    ref_iters = [int(rf['k%d%s_niter' % (k, t)]) for k in ks for t in ('p', 'm')]
    assert list(one['n_iter']) == ref_iters
    shards = []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        assert np.array_equal(d['rf'], one['rf']) and np.array_equal(d['I'], one['I'])
        assert list(d['n_iter']) == ref_iters
        shards.append(int(d['shard'][1]))
    assert sum(shards) == 6 and max(shards) - min(shards) <= 1 and (world != 8 or shards.count(0) == 2)


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
        # the per-iteration exchange of the multi-GPU bench: lsx_monitors leaves (dJ, dPops, NaN flag, singular flag) in a
        # device buffer on the engine's stream, RCCL reduces it in place, one read-back (MaxReducer.engine)
        from lightspinner_amd import fixtures, synth, Engine, _capi, drivers
        prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'))
        batch, _ = synth.perturbed_columns(prob, base, raw, ncol=5, seed=3, vlos_sigma=0.0)
        ts = torch.cuda.Stream()
        e1 = Engine(prob, 5, stream=ts.cuda_stream)
        e2 = Engine(prob, 5)
        for e in (e1, e2):
            e.set_columns(0, batch)
        red = MaxReducer(device=torch.device('cuda', 0), stream=ts)
        red.active = True
        for it in range(1, 6):
            a = drivers.mali_step(e1, it > 3, reducer=red)
            b = drivers.mali_step(e2, it > 3)
            assert a == b, (it, a, b)
        assert np.array_equal(e1.get(_capi.LSX_N), e2.get(_capi.LSX_N))
        # the same exchange split in two (MaxReducer.engine_begin / engine_end): reduction, all-reduce and read-back are enqueued,
        # the next formal solution goes out behind them, the host waits for the read-back only (drivers.mali_steps); and the
        # whole loop with the look-ahead formal solution taken back at the end (drivers.iterate_mali_engine)
        a = list(drivers.mali_steps(e1, 5, reducer=red, n_lambda_only=0, first=6, lookahead=True))
        b = [drivers.mali_step(e2, True) for _ in range(5)]
        assert a == b
        assert np.array_equal(e1.get(_capi.LSX_N), e2.get(_capi.LSX_N)) and np.array_equal(e1.get(_capi.LSX_I), e2.get(_capi.LSX_I))
        ha = drivers.iterate_mali_engine(e1, reducer=red, n_lambda_only=0, max_iter=40, pipelined=True)
        hb = drivers.iterate_mali_engine(e2, n_lambda_only=0, max_iter=40, pipelined=False)
        assert ha.converged and ha.dJ == hb.dJ and ha.dPops == hb.dPops
        for w in (_capi.LSX_N, _capi.LSX_I, _capi.LSX_J, _capi.LSX_GAMMA):
            assert np.array_equal(e1.get(w), e2.get(w))
        e3 = Engine(prob, 1, stream=ts.cuda_stream)            # singular system (Gamma still zero) -> LinAlgError on every rank
        e3.set_columns(0, base)
        e3.stat_equil_async()
        with pytest.raises(np.linalg.LinAlgError):
            red.engine(e3)
    finally:
        dist.destroy_process_group()
