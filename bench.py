#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MALI formal-solution hot path on MI355X.

A "step" = one MALI iteration (formal_sol_gamma_matrices + stat_equil, rh_method.py:565,710)
over every atmosphere column resident on the GPU.  Workload (config.workload):
  c3 (default)  1000 FALC-perturbed CaII columns per GPU, ray-dependent line profiles
                (BASELINE.json configs[2]; weak scaling: columns per GPU fixed)
  c2            the single FALC CaII column (configs[1]; 45 wavefronts -- latency bound by
                construction, reported inside every run as `falc_single_column`)
  c4            FALC-perturbed Ca+H columns (configs[3] per-GPU share, 1250 columns); its one-GPU share is
                also measured inside every default run (`c4_share`)
  c5            the CaII temperature response function (configs[4]): 164 perturbed columns, sharded over the ranks
value = depth-points x wavelengths x rays updated per second, whole job.
Inputs are resident in HBM before the timed region.  One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# hardware queues of the HIP runtime (read when it initialises; lightspinner_amd/__init__.py says why): set before anything imports
# torch, recorded in the JSON line (config.runtime_env)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (the spec peak the roofline fraction is priced against)
HBM_MEASURED_GBPS = 6290.0        # the guide's measured achievable streaming rate (reported beside it: frac_of_measured_peak)
N_SIMD, CLOCK_HZ = 1024, 2.4e9    # 256 CUs x 4 SIMD-32, max clock
HELD_CLOCK_HZ = 1.9e9             # the clock the chip holds under the ray-serial sweep (in-kernel s_memtime / s_memrealtime: 1.85-1.92 GHz
                                  # on C3, 2.03-2.13 on C4; profiles/r03_bound_evidence.md 7)
VALU_PEAK = N_SIMD * CLOCK_HZ / 4.0   # wave64 fp64 VALU instructions per second (4 cycles each)


def generate_columns(path, first, ncol, compact, seed=1234):
    """synthetic ensemble columns [first, first+ncol) -> (ColumnBlock, profile inputs or None); with a line-of-sight velocity
    (not compact) the profiles are left to the library (lsx_set_line_profiles builds them on the device)"""
    from lightspinner_amd import fixtures, synth
    prob, block, raw = fixtures.load_problem_npz(path, phi_compact=compact)
    return synth.perturbed_columns(prob, block, raw, ncol=ncol, seed=seed, first=first, vlos_sigma=0.0 if compact else 2.0e3)


def load_columns(eng, batch, prof, c0=0, step=100):
    """inputs -> engine: arrays by lsx_set_columns, line profiles by lsx_set_line_profiles where they were not handed over"""
    from lightspinner_amd import synth
    synth.load_columns(eng, batch, prof, col0=c0, step=step)


_REAL_STDOUT = None


def protect_stdout():
    """stdout carries ONE JSON line: whatever libraries print to file descriptor 1 during the run (gloo's rank chatter in a
    rehearsal, runtime notices) is sent to stderr instead; emit() writes the line to the real stdout"""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)


def emit(result):
    out = _REAL_STDOUT or sys.stdout
    out.write(json.dumps(result) + '\n')
    out.flush()


def stats_ms(times):
    t = np.asarray(times) * 1e3
    return dict(min=float(t.min()), median=float(np.median(t)), max=float(t.max()), p10=float(np.percentile(t, 10)),
                p90=float(np.percentile(t, 90)), n=int(t.size))


def dist_env():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init_gpu(local_rank, world):
    """-> (torch, dist, device for collectives, rehearsal info or None).  LSX_BENCH_REHEARSE=1 runs the N > 1 flow with
    every rank on GPU 0 and gloo instead of RCCL (RCCL refuses two ranks on one device): a rehearsal of the multi-rank code
    path on a one-GPU box, labelled as such in the output -- never a multi-GPU measurement."""
    import torch
    import torch.distributed as dist
    rehearse = os.environ.get('LSX_BENCH_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    dev = torch.device('cpu') if rehearse else torch.device('cuda', local_rank)
    info = dict(rehearsal=True, backend='gloo', gpus_physical=1,
                note='all ranks share ONE GPU and exchange over gloo: exercises the multi-rank code path only; the value is '
                     'NOT a multi-GPU measurement') if rehearse else None
    return torch, dist, dev, local_rank, info


def parallelism_text(world, rehearsal):
    if world == 1:
        return 'one GPU (columns are independent 1-D problems; no collective)'
    if rehearsal:
        return 'REHEARSAL: %d ranks on one GPU, gloo all-reduce(MAX) of the convergence monitors per iteration' % world
    return 'columns sharded over %d GPUs, no data-path collective; RCCL all-reduce(MAX) of (dJ, dPops, flags) per iteration' % world


def run_c5(args, rank, local_rank, world):
    """C5 (response_fn.py): T[k] +- 25 K at every depth -> 164 perturbed FALC CaII columns, warm started from the
    converged base column, every column iterated to ITS OWN convergence; rf = (I+ - I-) / I_base.  Over several ranks
    the perturbed columns are block partitioned (164 -> 21 / 20 per GPU at 8), all_done = AND over ranks, the
    intensities are gathered at the end.  Inputs: tests/golden/rf_ca_inputs.npz (all depths, written by make_golden.py
    rf_inputs); the three depths of rf_ca.npz carry the reference's converged intensities and pin the result."""
    from lightspinner_amd import fixtures, response, _capi
    from lightspinner_amd.parallel import AllDone
    gold = os.path.join(ROOT, 'tests', 'golden')
    prob, base, raw = fixtures.load_problem_npz(os.path.join(gold, 'falc_ca.npz'))
    fx = dict(np.load(os.path.join(gold, 'rf_ca_inputs.npz')))
    ks = list(range(int(fx['Nspace'])))
    torch, dist, dev, local_rank, rehearsal = init_gpu(local_rank, world)
    lib = _capi.load_hip_library()
    kw = dict(lib=lib, device=local_rank, rank=rank, world=world)
    if world > 1:
        kw.update(all_done=AllDone(), gather=response.gather_over_ranks())
    response.run_response_function(prob, base, fx, ks[:2 * world], **kw)          # warm-up (library load, first launches)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = response.run_response_function(prob, base, fx, ks, **kw)      # ends with a blocking read-back / gather of I
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    scaling_record = None
    if world > 1:
        dt_own = dt
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        every = [None] * world
        dist.all_gather_object(every, dict(rank=rank, seconds=dt_own, world_seen=dist.get_world_size(), backend=dist.get_backend(),
                                           shard=[int(x) for x in out['shard']]))
        scaling_record = dict(world_seen=dist.get_world_size(), backend=dist.get_backend(), per_rank=every, seconds_max_over_ranks=dt)
    ref = dict(np.load(os.path.join(gold, 'rf_ca.npz')))
    err = 0.0
    for k in [int(k) for k in ref['ks']]:
        r = (ref['k%dp_I' % k][:, -1] - ref['k%dm_I' % k][:, -1]) / ref['base_I'][:, -1]
        err = max(err, float(np.max(np.abs(out['rf'][:, k] - r)) / np.max(np.abs(r))))
    col_iters = int(out['n_iter'].sum()) + out['n_iter_base'] * world     # every rank solves the base column itself
    units = prob.work_units_per_column() * col_iters
    steps = int(out['n_iter'].max()) + out['n_iter_base']
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    result = dict(
        metric='depth_points_x_wavelengths_x_rays_per_sec', value=units / dt, unit='point-updates/s', n_gpus=world, steps=steps,
        warmup=0, ms_per_step=dt / steps * 1e3, higher_is_better=True, scaling='strong', vs_baseline=None, dtype='f64',
        data='FALC CaII + reference-generated temperature perturbations (tests/golden/rf_ca_inputs.npz)',
        config=dict(workload='C5: CaII temperature response function, %d perturbed columns (T[k] +- 25 K at %d depths) + base, '
                             'per-column convergence' % (2 * len(ks), len(ks)), columns_total=2 * len(ks) + 1,
                    Nspace=prob.Nspace, Nspect=prob.Nspect, Nrays=prob.Nrays, parallelism=parallelism_text(world, rehearsal)),
        response_function=dict(seconds_total=dt, base_iterations=out['n_iter_base'],
                               perturbed_iterations_min=int(out['n_iter'].min()), perturbed_iterations_max=int(out['n_iter'].max()),
                               column_iterations=col_iters, rf_shape=list(out['rf'].shape),
                               max_abs_err_vs_reference_rf_over_max=err,
                               reference_depths_checked=[int(k) for k in ref['ks']],
                               note='the reference runs these 165 MALI solves one after the other in pure Python (~2 h here)'),
        roofline=None, cpu_baseline=None)
    if scaling_record is not None:
        result['scaling_record'] = scaling_record
    if rehearsal:
        result.update(rehearsal)
    emit(result)


def cpu_baseline(prob, batch, prof, seconds_target=12.0):
    """The oracle (C restatement, kind 'port') on a bounded sample of the same workload, on this
    host's cores, OpenMP over columns.  Checker/baseline only -- never the product path."""
    import oracle
    from lightspinner_amd import Engine
    lib = oracle.load()
    # the GPU box gives one GPU's share of the host: 16 cores (more threads than that only thrash)
    cores = min(os.cpu_count() or 1, int(os.environ.get('LSX_CPU_THREADS', '16')))
    nsample = min(batch.ncol, 4 * cores)
    eng = Engine(prob, nsample, lib=lib)
    load_columns(eng, batch.slice(0, nsample), None if prof is None else tuple(None if p is None else p[:nsample] for p in prof))
    # one thread on the SINGLE FALC CaII column (the reference's own case, test.py), beside the batch figures
    single = None
    try:
        from lightspinner_amd import fixtures
        p1, b1, _ = fixtures.load_problem_npz(os.path.join(ROOT, 'tests', 'golden', 'falc_ca.npz'))
        e1 = Engine(p1, 1, lib=lib)
        e1.set_columns(0, b1)
        lib.check(lib.dll.lsx_oracle_set_threads(e1._h, 1))
        e1.formal_sol_gamma()
        t0 = time.perf_counter()
        n1 = 0
        while time.perf_counter() - t0 < 1.5:
            e1.formal_sol_gamma(); e1.stat_equil()
            n1 += 1
        dt1 = time.perf_counter() - t0
        e1.close()
        single = dict(value=p1.work_units_per_column() * n1 / dt1, mali_iters_per_sec=n1 / dt1, iters=n1, seconds=dt1, threads=1,
                      sample='the single FALC CaII column (82 depth x 287 wavelengths x 5 rays x 2), FS + SE per iteration, one thread')
    except Exception as e:                       # the batch figures below do not depend on it
        single = dict(error=str(e))
    out = {}
    for label, nthreads in (('1thread', 1), ('allcores', cores)):
        lib.check(lib.dll.lsx_oracle_set_threads(eng._h, nthreads))
        # calibrate with one iteration, then run ~seconds_target/2 each
        t0 = time.perf_counter()
        eng.formal_sol_gamma(); eng.stat_equil()
        t1 = time.perf_counter() - t0
        iters = int(max(1, min(200, (seconds_target / 2) / max(t1, 1e-6))))
        t0 = time.perf_counter()
        for _ in range(iters):
            eng.formal_sol_gamma()
            eng.stat_equil()
        dt = time.perf_counter() - t0
        out[label] = dict(value=prob.work_units_per_column() * nsample * iters / dt, iters=iters, seconds=dt,
                          threads=nthreads)
    eng.close()
    return dict(value=out['allcores']['value'], unit='point-updates/s', cores=cores, kind='port',
                sample='%d columns of the same workload x %d MALI iterations (FS+SE), OpenMP over columns; '
                       'single thread: %.4g point-updates/s' % (nsample, out['allcores']['iters'], out['1thread']['value']),
                value_1thread=out['1thread']['value'], host_cpu_count=os.cpu_count(),
                cores_note='threads used = min(os.cpu_count(), LSX_CPU_THREADS or 16): the GPU box gives one GPU 16 cores of its host',
                single_column_1thread=single)


def profile_figures(workload):
    """counter-derived figures of an EARLIER, builder-run rocprofv3 collection (profiles/pmc_figures.json, written by
    profiles/summarize.py): HBM bytes and VALU instructions per column per formal-solution call, with the hash of the kernel
    sources they were collected from.  -> (figures or None, file name, stale: the sources beside this library differ)"""
    path = os.path.join(ROOT, 'profiles', 'pmc_figures.json')
    try:
        fig = json.load(open(path)).get(workload)
    except Exception:
        return None, None, None
    if not fig:
        return None, None, None
    sys.path.insert(0, os.path.join(ROOT, 'profiles'))
    import srchash
    return fig, 'profiles/pmc_figures.json', fig.get('csrc_hash') != srchash.csrc_hash()


def roofline_block(eng, lib, prob, ncol, workload, kernel_reps):
    """roofline of the dominant kernel, measured live with HIP events on the kernels' own streams (lsx_time_formal_sol)"""
    import ctypes as C
    ms_total, ms_sweep = eng.time_formal_sol(2, kernel_reps)
    lib.dll.lsx_hip_info.argtypes = [C.c_void_p, C.c_int32]
    lib.dll.lsx_hip_info.restype = C.c_double
    info = lambda w: float(lib.dll.lsx_hip_info(eng._h, w))
    balg = eng.algorithmic_bytes_per_column()      # SURVEY 8d formula, whole FS call
    bsweep = info(0)                               # the part of it the sweep kernel itself moves
    ach = bsweep * ncol / (ms_sweep * 1e-3) / 1e9
    fig, src, stale = profile_figures(workload)
    traffic = valu = traffic_all = None
    if fig and not stale:
        if fig.get('hbm_bytes_per_call_per_column') is not None:
            traffic = fig['hbm_bytes_per_call_per_column'] * ncol
        if fig.get('hbm_bytes_per_call_per_column_all_kernels') is not None:      # every kernel of the call, not the sweep classes only
            traffic_all = fig['hbm_bytes_per_call_per_column_all_kernels'] * ncol
        if fig.get('valu_insts_per_call_per_column') is not None:
            rate = fig['valu_insts_per_call_per_column'] * ncol / (ms_sweep * 1e-3)
            valu = dict(achieved=rate, peak=VALU_PEAK, unit='wave64 VALU instructions/s', frac=rate / VALU_PEAK,
                        frac_at_held_clock=rate / (N_SIMD * HELD_CLOCK_HZ / 4.0), held_clock_GHz=HELD_CLOCK_HZ / 1e9,
                        peak_definition='peak / frac: 1024 SIMDs x 2.4 GHz / 4 cycles, the issue rate of fp64 VALU instructions at the MAXIMUM clock; '
                                        'frac_at_held_clock: the same at the 1.9 GHz the chip holds under this kernel (in-kernel s_memtime / '
                                        's_memrealtime: 1.85-1.92 GHz on C3, 2.03-2.13 on C4; profiles/r03_bound_evidence.md 7)',
                        f64_share_of_valu_insts=fig.get('f64_share_of_valu_insts'),
                        insts_per_launch=fig['valu_insts_per_call_per_column'] * ncol,
                        source='%s (instruction counts from a builder-run rocprofv3 --pmc pass of this command at source hash %s, per '
                               'column; divided by the LIVE duration)' % (src, fig.get('csrc_hash')))
    # ---- ceiling model: what the bytes the call REALLY moves and the vector instructions it REALLY issues
    # allow at best, t_min = max(counter bytes / 6.29 TB/s, VALU instructions x 4 cycles / (1024 SIMDs x held clock)), against the live
    # durations -- per call, for the sweeps, and per kernel alone (the profile run's serialized durations).  `bound` is the larger term.
    ceiling = None
    bound = 'hbm'
    if fig and not stale and traffic_all and fig.get('valu_insts_per_call_per_column_all_kernels'):
        bw, held = fig.get('achievable_bw_Bps', HBM_MEASURED_GBPS * 1e9), fig.get('held_clock_Hz', HELD_CLOCK_HZ)
        price = 4.0 / (N_SIMD * held)
        tb_call, tv_call = traffic_all / bw * 1e3, fig['valu_insts_per_call_per_column_all_kernels'] * ncol * price * 1e3
        tb_sw, tv_sw = traffic / bw * 1e3, fig['valu_insts_per_call_per_column'] * ncol * price * 1e3
        bound = 'hbm' if tb_call >= tv_call else 'valu'
        per_kernel = None
        if fig.get('ceiling_per_kernel'):
            per_kernel = {k: dict(t_bytes_ms=v['hbm_bytes_per_column'] * ncol / bw * 1e3, t_valu_ms=v['valu_insts_per_column'] * ncol * price * 1e3,
                                  bound=v['bound'], alone_ms_profiled=v['alone_ms'], frac_of_ceiling_profiled=v['frac_of_ceiling'])
                          for k, v in fig['ceiling_per_kernel'].items()}
        ceiling = dict(achievable_bw_GBps=bw / 1e9, held_clock_GHz=held / 1e9,
                       call=dict(t_bytes_ms=tb_call, t_valu_ms=tv_call, t_min_ms=max(tb_call, tv_call), bound=bound, measured_ms=ms_total,
                                 frac_of_ceiling=max(tb_call, tv_call) / ms_total),
                       sweeps=dict(t_bytes_ms=tb_sw, t_valu_ms=tv_sw, t_min_ms=max(tb_sw, tv_sw), bound='hbm' if tb_sw >= tv_sw else 'valu',
                                   measured_ms=ms_sweep, frac_of_ceiling=max(tb_sw, tv_sw) / ms_sweep),
                       per_kernel_alone=per_kernel,
                       definition='t_min = max(HBM bytes from the counters / achievable bandwidth, wave64 VALU instructions x 4 cycles / (1024 SIMDs x '
                                  'the clock the chip holds under the kernel)); frac_of_ceiling = t_min / measured (live HIP events for `call` and '
                                  '`sweeps`; per_kernel_alone: each kernel alone on the machine in the builder\'s counter run at profiled_columns = %s, '
                                  'profiles/summarize.py).  Bytes and instruction counts: %s at source hash %s, per column x columns; 6.29 TB/s and the '
                                  'held clock: MI355X_MICROARCH.md / profiles/r03_bound_evidence.md 7.  A call whose bytes AND instructions overlapped '
                                  'perfectly would take t_min: frac_of_ceiling says how far the kernels are from what the work they do allows, `frac` '
                                  '(algorithmic bytes / 8 TB/s) how far the DESIGN is from the algorithm\'s bytes.'
                                  % (fig.get('ceiling_profiled_columns'), src, fig.get('csrc_hash')))
    ach_call = balg * ncol / (ms_total * 1e-3) / 1e9
    return dict(bound=bound, bound_means='the larger term of the ceiling model (`ceiling`: counter bytes / 6.29 TB/s against VALU instructions / issue rate at the held clock); '
                                         '"hbm" when the counters are stale (the roofline the metric is defined on, BASELINE.json north_star)',
                binding='two floors, neither reached: the HBM bytes the call really moves (2.3 x the algorithmic bytes on C4, 1.5 x on C3; 0.73-0.75 of the measured '
                        'call at 6.29 TB/s) and vector issue (0.51-0.57), overlapped as far as two waves per SIMD allow; what a wave waits for is the latency of its own '
                        'requests and dependent instructions (ablation, profiles/r06_bound_evidence.md 5: both directions SHARING their ray-independent reads, -12 % of '
                        'the sweeps\' bytes, buys 0.6 %)',
                kernel='one formal solution = every kernel of the call (SURVEY 8d: Ncol B_alg / t_FS): operand-table build, the sweep classes side by side '
                       '(lsx_sweep_rs_kernel<slots,lines,linked,topo,...>, ray-serial, five columns per wavefront: every tile class with at most two per-ray '
                       'slots in contexts of >= 160 columns; lsx_sweep_kernel<...>, one ray per lane: the other classes and smaller contexts), the '
                       'fast-continuum epilogue behind each class, the Gamma epilogue.  The sweep classes alone: `sweep_*`',
                # headline: the SURVEY 8d definition -- algorithmic bytes of the whole call over the whole call's duration
                achieved=ach_call, peak=HBM_PEAK_GBPS, unit='GB/s', frac=ach_call / HBM_PEAK_GBPS, frac_of_measured_peak=ach_call / HBM_MEASURED_GBPS,
                avg_launch_ms=ms_total, alg_bytes_per_launch=balg * ncol,
                # ... and the dominant kernels alone (rounds 1-5's headline): the sweep's share of the algorithmic bytes over the sweep span
                sweep_achieved=ach, sweep_frac=ach / HBM_PEAK_GBPS, sweep_frac_of_measured_peak=ach / HBM_MEASURED_GBPS, sweep_span_ms=ms_sweep,
                sweep_alg_bytes_per_launch=bsweep * ncol,
                measured_peak=HBM_MEASURED_GBPS, traffic=traffic_all, sweep_traffic=traffic,
                # the same durations against the bytes REALLY moved past the L2 (counter figure, over-fetch and the second
                # direction's re-reads included): how close the kernels are to the memory system in real, not algorithmic, bytes
                traffic_GBps=(traffic_all / (ms_total * 1e-3) / 1e9) if traffic_all else None,
                sweep_traffic_GBps=(traffic / (ms_sweep * 1e-3) / 1e9) if traffic else None,
                traffic_frac_of_peak=(traffic_all / (ms_total * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic_all else None,
                traffic_frac_of_measured_peak=(traffic_all / (ms_total * 1e-3) / 1e9 / HBM_MEASURED_GBPS) if traffic_all else None,
                traffic_over_alg=(traffic_all / (balg * ncol)) if traffic_all else None,
                traffic_stale=bool(stale) if fig else None,
                traffic_source=('%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a builder run at source hash %s, (2*FETCH_SIZE + '
                                'WRITE_SIZE)*1024 per call per column x columns, every kernel of the call (`sweep_traffic`: the sweep classes); not measured in this run%s'
                                % (src, fig.get('csrc_hash'), '; STALE: the kernel sources have changed since, so the figure is withheld' if stale else ''))
                if fig else None,
                ceiling=ceiling, valu=valu,
                limited_by='latency at two waves per SIMD between two floors (`ceiling`: bytes first, vector issue second). A C4 call moves 2.3 x its algorithmic bytes -- both directions of a '
                           'column read the ray-independent streams (they visit the depths in opposite order: the reuse distance is a workgroup\'s lifetime, '
                           'far beyond the L2: cache-policy hints change FETCH_SIZE by 1-2 %, profiles/r06_bound_evidence.md), J leaves as a half and again '
                           'as a total, the angle sums of Psi* leave per direction for the fast-continuum epilogue -- at 4.1-4.5 TB/s of the 6.29 achievable; '
                           'the sweeps hold 150-256 vector registers (two waves per SIMD) and issue 0.7 of what the byte floor\'s time allows; removing the second reads altogether '
                           '(ablation) changes the call by 0.6 %: the bytes are a floor, not what the waves wait for. History of what '
                           'was tried and measured: profiles/r03..r06_bound_evidence.md',
                fs_call=dict(ms=ms_total, ms_sweep_kernel=ms_sweep, ms_gamma_finish=info(4), ms_fast_epilogue_exposed=info(6),
                             ms_note='ms_sweep_kernel: fork -> end of the last class\'s sweep; ms_fast_epilogue_exposed: from there to the join (the fast-continuum '
                                     'epilogue of the class that finishes last, which nothing hides); ms_gamma_finish: the Gamma epilogue behind the join; ms: '
                                     'everything, operand-table build included',
                             alg_bytes=balg * ncol,
                             traffic=traffic_all, traffic_over_alg=(traffic_all / (balg * ncol)) if traffic_all else None,
                             traffic_note='HBM bytes of EVERY kernel of a formal solution (sweeps, fast-continuum kernels, operand-table build, '
                                          'Gamma epilogue) from the same counter passes as roofline.traffic (profiles/summarize.py: '
                                          'fs_call_hbm_bytes_per_call); null when those figures are stale',
                             achieved_GBps=ach_call, frac=ach_call / HBM_PEAK_GBPS, frac_of_measured_peak=ach_call / HBM_MEASURED_GBPS),
                point_updates_per_sec_kernel=prob.work_units_per_column() * ncol / (ms_sweep * 1e-3),
                tiles_per_column=info(1), wavelengths_per_tile=info(3), lds_bytes_per_workgroup=info(2),
                slab_bytes_per_column=info(5),
                note='achieved / peak / frac = ALGORITHMIC bytes of one formal solution (SURVEY 8d: Ncol B_alg) over the HIP-event duration of the whole '
                     'call (events on the context\'s stream around the call: every kernel, the forked class streams joined back), against the 8 TB/s HBM '
                     'peak the metric is defined on.  sweep_*: the sweep classes alone (their share of the bytes over fork -> last class end).  `ceiling` '
                     'says what the kernels run against.')


def timed_steps(eng, reducer, nsteps, warmup, barrier):
    from lightspinner_amd import drivers
    # drivers.mali_steps: the MALI loop as the product runs it -- where the library says it pays (lsx_prefers_lookahead: small
    # contexts) the next iteration's formal solution is enqueued while the host waits for the monitors of the current one, else
    # the plain sequence; either way exactly `nsteps` formal solutions and `nsteps` stat_equil calls inside the timed region and
    # nothing enqueued beyond the last step
    for _ in drivers.mali_steps(eng, warmup, reducer):
        pass
    barrier()
    per = []
    t0 = ta = time.perf_counter()
    dJ = dP = None
    for dJ, dP in drivers.mali_steps(eng, nsteps, reducer):
        tb = time.perf_counter()
        per.append(tb - ta)
        ta = tb
    barrier()
    return time.perf_counter() - t0, per, dJ, dP


def measure_share(workload, ncol, local_rank, lib, torch, steps=40, warmup=3, kernel_reps=10):
    """one GPU's share of another configuration, measured in the same run (used for C4: the north-star workload's
    per-GPU share of 1250 Ca+H columns) -> dict"""
    from lightspinner_amd import fixtures, Engine
    fixture = os.path.join(ROOT, 'tests', 'golden', 'falc_cah.npz' if workload == 'c4' else 'falc_ca.npz')
    prob, base, raw = fixtures.load_problem_npz(fixture, phi_compact=False)
    t0 = time.time()
    batch, prof = generate_columns(fixture, 0, ncol, False)
    t_gen = time.time() - t0
    ts = torch.cuda.Stream()
    eng = Engine(prob, ncol, device=local_rank, stream=ts.cuda_stream, lib=lib)
    load_columns(eng, batch, prof)
    del batch
    policy = eng.sweep_policy()
    dt, per, dJ, dP = timed_steps(eng, None, steps, warmup, torch.cuda.synchronize)
    units = prob.work_units_per_column() * ncol * steps
    roof = roofline_block(eng, lib, prob, ncol, workload, kernel_reps)
    eng.close()
    return dict(workload='C4 share: %d FALC-perturbed Ca+H columns on one GPU (10 000 / 8), 82 depth x %d wavelengths x 5 rays x 2'
                         % (ncol, prob.Nspect),
                columns=ncol, steps=steps, value=units / dt, unit='point-updates/s', ms_per_step=dt / steps * 1e3,
                step_ms=stats_ms(per), last_dJ=dJ, last_dPops=dP, sweep_span_ms=roof['sweep_span_ms'], roofline=roof,
                generate_s=t_gen, sweep_policy=policy)


def dry_run(args, rank, world):
    """what every rank does around the timed region, without the GPU: rendezvous, one all-reduce(MAX), one line from rank 0"""
    import torch
    import torch.distributed as dist
    top = rank
    if world > 1:
        dist.init_process_group('gloo')
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        top = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(dict(metric='depth_points_x_wavelengths_x_rays_per_sec', value=None, unit='point-updates/s', n_gpus=world,
                  steps=args.steps, warmup=args.warmup, dry_run=True, max_rank_seen=top,
                  note='launcher check only (--dry-run): nothing was measured'))


def self_launch(ngpus):
    """`python bench.py --gpus N` without a launcher: this process touches no GPU (no torch, no HIP library) and starts
    the N ranks as ONE child, `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`, rendezvous on
    127.0.0.1 at a free port; rank 0's JSON line is passed through, the exit status is the child's."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ngpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    sys.stdout.flush()
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in proc.stdout:              # the ranks send everything but the result to stderr (protect_stdout)
        if line.lstrip().startswith('{'):
            lines.append(line)
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    for line in lines[-1:]:
        sys.stdout.write(line)
    sys.stdout.flush()
    if rc != 0 or not lines:
        raise SystemExit(rc or 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='c3', choices=['c2', 'c3', 'c4', 'c5', 'all'])
    ap.add_argument('--columns', type=int, default=None, help='columns per GPU (default 1000 / 1 / 1250)')
    ap.add_argument('--compact-phi', action='store_true', help='vlos == 0: ray independent profiles (P = 1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--kernel-reps', type=int, default=20)
    ap.add_argument('--no-single-column', action='store_true',
                    help='skip the FALC single-column section (used for rocprofv3 runs so that every sweep launch has the workload size)')
    ap.add_argument('--no-c4-share', action='store_true', help='skip the C4 per-GPU share measured beside the default workload')
    ap.add_argument('--physical-columns', action='store_true',
                    help='columns whose populations, rates and profiles the library derives from each column\'s own perturbed '
                         'atmosphere (lsx_set_atmosphere, SURVEY 8f N1) instead of input-level perturbations')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher check without a GPU: the ranks rendezvous over gloo, all-reduce(MAX) their rank and rank 0 '
                         'prints a JSON line with value null')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)
    protect_stdout()
    rank, local_rank, world = dist_env()
    if args.dry_run:
        return dry_run(args, rank, world)
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` or under torch.distributed.run '
                         'with --nproc-per-node equal to --gpus' % (args.gpus, world))
    if args.workload == 'c5':
        return run_c5(args, rank, local_rank, world)
    fixture = os.path.join(ROOT, 'tests', 'golden', {'c4': 'falc_cah.npz', 'all': 'falc_all.npz'}.get(args.workload, 'falc_ca.npz'))
    ncol = args.columns or {'c2': 1, 'c3': 1000, 'c4': 1250, 'all': 250}[args.workload]
    compact = args.compact_phi or args.workload == 'c2'

    # ---- host-side input generation, before any GPU initialisation -------------------
    from lightspinner_amd import fixtures, Engine, _capi, drivers
    from lightspinner_amd.parallel import MaxReducer
    prob, base, raw = fixtures.load_problem_npz(fixture, phi_compact=compact)
    t0 = time.time()
    batch, prof = generate_columns(fixture, rank * ncol, ncol, compact) if ncol > 1 else (base, None)
    t_gen = time.time() - t0

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # CPU baseline: rank 0 at N = 1 only
        cpu = cpu_baseline(prob, batch, prof)

    # ---- GPU ---------------------------------------------------------------------------
    torch, dist, dev, local_rank, rehearsal = init_gpu(local_rank, world)
    lib = _capi.load_hip_library()
    ts = torch.cuda.Stream()                      # the engine launches on this stream; so does the collective (MaxReducer)
    # the sweep kernel is chosen for the WHOLE problem (all ranks' columns), not for this rank's shard of it (lsx_set_sweep_policy)
    eng = Engine(prob, ncol, device=local_rank, stream=ts.cuda_stream, lib=lib, policy_columns=ncol * world)
    t0 = time.time()
    load_columns(eng, batch, prof)
    t_up = time.time() - t0
    t_chain = None
    if args.physical_columns and ncol > 1:
        # the same ensemble, made physically consistent: broadening, damping, LTE populations, collisional rates and line
        # profiles from each column's own atmosphere, on the device (outside the timed region, like every set-up step)
        from lightspinner_amd import atomdata, synth
        sd = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'setup_falc.npz')))
        names = [str(x) for x in sd['atom_names']]
        eng.set_atomic_data(atomdata.from_fixture(sd, atoms=[names.index(n) for n in prob.atom_names]))
        atm = synth.perturbed_atmospheres(prob, raw, ncol, first=rank * ncol, vlos_sigma=0.0 if compact else 2.0e3)
        t0 = time.time()
        eng.set_atmosphere(0, lte_pops=True, **atm)
        torch.cuda.synchronize()
        t_chain = time.time() - t0
    upload_bytes = sum(getattr(batch, k).nbytes for k in ('phi', 'bg_chi', 'bg_eta', 'C', 'n', 'nStar') if getattr(batch, k) is not None)
    if prof is not None:
        upload_bytes += sum(p.nbytes for p in prof if p is not None)
    # one MALI iteration: both calls are enqueued, the monitors are read once -- what the reference's
    # `while dJ > 2e-3 or dPops > 1e-3` loop needs per iteration (test.py:23-29).  Over several ranks the monitors stay
    # on the device until RCCL has reduced them (parallel.MaxReducer.engine)
    reducer = MaxReducer(device=dev, stream=ts).attach(eng) if world > 1 else None      # (attach: the ranks compare their engines' options signatures here, all at once)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt, per_step, dJ, dP = timed_steps(eng, reducer, args.steps, args.warmup, barrier)
    # the scaling record proves itself (round 5): what every rank saw -- its own time for the K steps, the world size and backend of
    # the process group it was in, the sweep mapping its engine ran and its options signature -- gathered onto rank 0's line
    scaling_record = None
    if world > 1:
        dt_own = dt
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = dict(rank=rank, ms_per_step=dt_own / args.steps * 1e3, world_seen=dist.get_world_size(), backend=dist.get_backend(),
                    sweep_policy=eng.sweep_policy(), columns=ncol, options_signature='%016x' % eng.options_signature(),
                    device=torch.cuda.get_device_name(dev) if dev.type == 'cuda' else str(dev))
        every = [None] * world
        dist.all_gather_object(every, mine)
        scaling_record = dict(world_seen=dist.get_world_size(), backend=dist.get_backend(), per_rank=every,
                              ms_per_step_max_over_ranks=dt / args.steps * 1e3)

    units = prob.work_units_per_column() * ncol * world * args.steps
    result = None
    if rank == 0:
        roofline = roofline_block(eng, lib, prob, ncol, args.workload, args.kernel_reps)

        # ---- parity + single-column (C2) numbers in the same run -----------------------
        single = None
        if not args.no_single_column:
            p1, b1, r1 = fixtures.load_problem_npz(os.path.join(ROOT, 'tests', 'golden', 'falc_ca.npz'))
            e1 = Engine(p1, 1, device=local_rank, stream=ts.cuda_stream, lib=lib)
            t_runs = []
            for _ in range(3):       # the first run pays for code objects and first launches; `seconds` is the best of the others
                e1.set_columns(0, b1)
                t0 = time.perf_counter()
                h = drivers.iterate_mali_engine(e1)
                t_runs.append(time.perf_counter() - t0)
            t_c2 = min(t_runs[1:])
            nref = fixtures.pops_from_raw(r1, 'conv', p1)
            n1 = e1.get(_capi.LSX_N)[0]
            single = dict(n_iter=h.n_iter, converged=h.converged, seconds=t_c2, seconds_first_run=t_runs[0], mali_iters_per_sec=h.n_iter / t_c2,
                          point_updates_per_sec=p1.work_units_per_column() * h.n_iter / t_c2,
                          max_dn_over_n_vs_ref=float(np.max(np.abs(n1 - nref) / np.abs(nref))),
                          max_dI_over_I_vs_ref=float(np.max(np.abs(e1.get(_capi.LSX_I)[0] - r1['conv_I']) / np.abs(r1['conv_I']))),
                          reference_python_iters_per_sec=0.9,
                          reference_python_note='0.9 it/s is NOT measured in this run: the reference\'s own Python on one core of the survey '
                                                'container, numba absent (SURVEY.md 6); the reference cannot travel to the GPU box',
                          us_per_mali_iteration=t_c2 / max(h.n_iter, 1) * 1e6)
            e1.close()
            # ... and what a user of the reference gets by changing one import (missing 1): the SAME loop, test.py:20-29,
            # through lightspinner_amd.rh_method.Context on reference-shaped objects (tests/helpers.build_fakes: the golden fixture's
            # atmosphere, spectrum, populations, background) -- two blocking calls per iteration, populations written back into the
            # caller's array after every stat_equil, I read once after the loop (inside the timed region)
            sys.path.insert(0, os.path.join(ROOT, 'tests'))
            from helpers import build_fakes
            from lightspinner_amd.rh_method import Context
            dgold = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'falc_ca.npz')))
            t_ctx, hc, n_ctx, I_ctx = [], None, None, None
            for _ in range(3):
                atmos, spect, eq, bg = build_fakes(dgold)
                ctx = Context(atmos, spect, eq, bg, device=local_rank, stream=ts.cuda_stream, lib=lib)
                t0 = time.perf_counter()
                hc = drivers.iterate_mali(ctx)
                I_ctx = np.array(ctx.I)
                t_ctx.append(time.perf_counter() - t0)
                n_ctx = np.array(eq['CA'].n)
                ctx.close()
            t_cx = min(t_ctx[1:])
            single['context_dropin'] = dict(
                n_iter=hc.n_iter, converged=hc.converged, seconds=t_cx, seconds_first_run=t_ctx[0], mali_iters_per_sec=hc.n_iter / t_cx,
                us_per_mali_iteration=t_cx / max(hc.n_iter, 1) * 1e6, ratio_to_engine_loop=t_cx / t_c2,
                max_dn_over_n_vs_ref=float(np.max(np.abs(n_ctx - nref) / np.abs(nref))),
                max_dI_over_I_vs_ref=float(np.max(np.abs(I_ctx - r1['conv_I']) / np.abs(r1['conv_I']))),
                readback='lazy (J, I, Gamma fetched when first looked at; n written back after every stat_equil)',
                loop='drivers.iterate_mali(Context) = test.py:20-29: formal_sol_gamma_matrices(); if i > 3: stat_equil(); one blocking call each')

        result = dict(metric='depth_points_x_wavelengths_x_rays_per_sec', value=units / dt, unit='point-updates/s',
                      n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
                      higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
                      config=dict(workload={'c2': 'C2: single FALC 82-depth CaII column, 5 rays',
                                            'c3': 'C3: %d FALC-perturbed CaII columns per GPU, 82 depth x 287 wavelengths x 5 rays x 2 directions' % ncol,
                                            'c4': 'C4 share: %d FALC-perturbed Ca+H columns per GPU, 82 depth x 777 wavelengths x 5 rays x 2' % ncol,
                                            'all': 'all five of the reference\'s model atoms active (H, C, MgII, CaII, Fe: 53 levels, 109 transitions): %d '
                                                   'FALC-perturbed columns per GPU, 82 depth x 3966 wavelengths x 5 rays x 2 (not a BASELINE configuration: '
                                                   'the largest problem the reference\'s own model atoms make)' % ncol}[args.workload],
                                  columns_per_gpu=ncol, columns_total=ncol * world, Nspace=prob.Nspace, Nspect=prob.Nspect,
                                  Nrays=prob.Nrays, profiles='compact (vlos=0)' if compact else 'ray dependent (vlos!=0)',
                                  columns='populations, rates and profiles derived by the library from each column\'s perturbed atmosphere' if t_chain is not None else 'input-level perturbations (BASELINE C3 / C4)',
                                  parallelism=parallelism_text(world, rehearsal), sweep_policy=eng.sweep_policy(),
                                  runtime_env=dict(GPU_MAX_HW_QUEUES=os.environ.get('GPU_MAX_HW_QUEUES'),
                                                   LSX_env={k: v for k, v in os.environ.items() if k.startswith('LSX_')},
                                                   effective_options=eng.effective_options(),
                                                   options_signature='%016x' % eng.options_signature())),
                      step_ms=dict(stats_ms(per_step), note='host clock on rank 0 between the monitor read-backs of consecutive steps; loop: ' +
                                   ('look-ahead (the next formal solution is enqueued before a read-back is waited for)' if eng.prefers_lookahead()
                                    else 'plain (formal solution, stat_equil, read-back; lsx_prefers_lookahead = 0 for this context)')),
                      mali_iters_per_sec=args.steps / dt, column_iters_per_sec=args.steps * ncol * world / dt,
                      last_dJ=dJ, last_dPops=dP,
                      roofline=roofline, cpu_baseline=cpu, falc_single_column=single,
                      max_dn_over_n_vs_ref=single['max_dn_over_n_vs_ref'] if single else None,
                      setup=dict(generate_s=t_gen, upload_s=t_up, upload_GB=upload_bytes / 1e9, device_chain_s=t_chain,
                                 line_profiles='built on the device (lsx_set_line_profiles)' if prof is not None else 'ray independent (vlos = 0): the fixture\'s profiles',
                                 note='PCIe-inclusive upload (and the device-side profile build) is outside the timed region (inputs resident in HBM)'))
        if scaling_record is not None:
            result['scaling_record'] = scaling_record
        if rehearsal:
            result.update(rehearsal)
    eng.close()
    del batch
    if rank == 0 and world == 1 and args.workload == 'c3' and not args.no_c4_share:
        result['c4_share'] = measure_share('c4', 1250, local_rank, lib, torch)
    if rank == 0:
        # short keys LAST, so that they survive a reader that keeps only the tail of the line: the headline workload and, at N = 1,
        # the north-star workload's per-GPU share (BASELINE.json: 10^4 Ca+H columns over 8 GPUs), each with its roofline fractions
        r = result['roofline']
        cl = r.get('ceiling') or {}
        result['summary'] = dict(workload=args.workload, ms_per_step=result['ms_per_step'], sweep_ms=r['sweep_span_ms'], sweep_frac=r['sweep_frac'],
                                 fs_call_ms=r['fs_call']['ms'], fs_call_frac=r['frac'], frac_of_measured_peak=r['frac_of_measured_peak'],
                                 traffic_over_alg=r['traffic_over_alg'], bound=r['bound'],
                                 frac_of_ceiling=(cl.get('call') or {}).get('frac_of_ceiling'))
        c4 = result.get('c4_share')
        if c4:
            q = c4['roofline']
            result['c4_ms_per_step'] = c4['ms_per_step']
            result['c4_updates_per_s'] = c4['value']
            result['c4_fs_call_ms'] = q['fs_call']['ms']
            result['c4_sweep_ms'] = q['sweep_span_ms']
            result['c4_sweep_frac'] = q['sweep_frac']
            result['c4_fs_call_frac'] = q['frac']
            result['c4_traffic_over_alg'] = q['traffic_over_alg']
            result['c4_frac_of_ceiling'] = ((q.get('ceiling') or {}).get('call') or {}).get('frac_of_ceiling')
        result['c3_sweep_frac' if args.workload == 'c3' else 'sweep_frac'] = r['sweep_frac']
        result['c3_fs_call_frac' if args.workload == 'c3' else 'fs_call_frac'] = r['frac']
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(result)


if __name__ == '__main__':
    main()
