"""Driver loops of the reference, over an Engine / Context with many columns.

D1  test.py:20-29 / response_fn.py:11-21  -- MALI iteration to convergence
D2  response_fn.py:23-67                  -- brute-force temperature response function
"""
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np


@dataclass
class MaliHistory:
    dJ: List[float] = field(default_factory=list)
    dPops: List[float] = field(default_factory=list)   # nan while only J is iterated
    converged: bool = False
    nonfinite: bool = False     # some monitor was NaN / inf on the way (the reference carries on silently)

    @property
    def n_iter(self):
        return len(self.dJ)


def iterate_mali(ctx, dJ_tol=2e-3, dPops_tol=1e-3, n_lambda_only=3, max_iter=500,
                 reduce_max: Optional[Callable[[float, float], tuple]] = None, log=None) -> MaliHistory:
    """`while dJ > 2e-3 or dPops > 1e-3` loop of test.py:20-29 with the reference's
    defaults: the first 3 iterations update J only (`if i > 3`, test.py:27).

    ctx needs formal_sol_gamma_matrices() and stat_equil() (Context or
    an Engine adapter).  reduce_max((dJ, dPops)) -> (dJ, dPops) is the hook where the
    multi-GPU driver takes the max over ranks (parallel.allreduce_max)."""
    h = MaliHistory()
    dJ, dPops, i = 1.0, 1.0, 0
    while dJ > dJ_tol or dPops > dPops_tol:
        i += 1
        dJ = ctx.formal_sol_gamma_matrices()
        if i > n_lambda_only:
            dPops = ctx.stat_equil()
        if reduce_max is not None:
            dJ, dPops = reduce_max(dJ, dPops)
        h.dJ.append(dJ)
        h.dPops.append(dPops if i > n_lambda_only else float('nan'))
        if log:
            log('Iteration %.3d: dJ: %.2e, dPops: %s' % (i, dJ, 'Just iterating Jbar' if i <= n_lambda_only else '%.2e' % dPops))
        if not (np.isfinite(dJ) and np.isfinite(dPops)):
            h.nonfinite = True          # the loop condition decides, exactly as in the reference: NaN > tol is False
        if i >= max_iter:
            break
    h.converged = (dJ <= dJ_tol and dPops <= dPops_tol)
    return h


def mali_step(engine, with_stat_equil=True, reducer=None):
    """One MALI iteration over every column of `engine` (problem.Engine): both calls are enqueued, then the
    convergence monitors are read ONCE -- lsx_sync on one rank, or reducer.engine(engine) over several (parallel.
    MaxReducer: the monitors stay on the device until the all-reduce has run).  -> (dJ, dPops or None)"""
    engine.formal_sol_gamma_async()
    if with_stat_equil:
        engine.stat_equil_async()
    if reducer is not None:
        dJ, dP = reducer.engine(engine)
    else:
        dJ, dP = engine.sync()
    return dJ, (dP if with_stat_equil else None)


def _begin(engine, reducer):
    if reducer is not None:
        reducer.engine_begin(engine)
    else:
        engine.sync_begin()


def _end(engine, reducer):
    return reducer.engine_end(engine) if reducer is not None else engine.sync_end()


def mali_steps(engine, nsteps, reducer=None, n_lambda_only=0, first=1, lookahead=None):
    """`nsteps` MALI iterations, yielding (dJ, dPops or None) after each, WITHOUT a host round trip between them: the next
    iteration's formal solution is enqueued before the monitors of the current one are waited for (include/lsx.h, lsx_sync_begin /
    lsx_sync_end), so the GPU goes from one iteration into the next while the host reads.  The iterations are numbered from
    `first`; those with a number > n_lambda_only include stat_equil (test.py:27).  The caller consumes every step (no early
    exit: nothing is rolled back here -- iterate_mali_engine does that).  lookahead (default: lsx_prefers_lookahead) False: the plain
    sequence of mali_step calls."""
    if nsteps < 1:
        return
    if lookahead is None:
        lookahead = engine.prefers_lookahead()
    if not lookahead:
        for s in range(nsteps):
            yield mali_step(engine, first + s > n_lambda_only, reducer)
        return
    i = first
    engine.formal_sol_gamma_async()
    se = i > n_lambda_only
    if se:
        engine.stat_equil_async()
    _begin(engine, reducer)
    for s in range(nsteps):
        last = s == nsteps - 1
        if not last:
            engine.formal_sol_gamma_async()            # iteration i + 1: nothing to take back, so the plain call
        dJ, dP = _end(engine, reducer)
        yield dJ, (dP if se else None)
        if not last:
            i += 1
            se = i > n_lambda_only
            if se:
                engine.stat_equil_async()
            _begin(engine, reducer)


def iterate_mali_engine(engine, reducer=None, dJ_tol=2e-3, dPops_tol=1e-3, n_lambda_only=3, max_iter=500, log=None,
                        pipelined=None) -> MaliHistory:
    """iterate_mali (test.py:20-29) on an Engine with many columns, globally converged: the loop runs until the maxima
    over all columns (and, with a reducer, over all ranks) are below the thresholds.

    pipelined (default: where the library says it pays, lsx_prefers_lookahead -- small contexts, e.g. a single column): iteration
    i + 1's formal solution is enqueued speculatively before the monitors of iteration i are known and discarded if the loop ends
    there (lsx_formal_sol_gamma_speculative / lsx_discard_formal_sol) -- the same iterations, the same results bit for bit, no
    idle GPU between iterations."""
    h = MaliHistory()
    if pipelined is None:
        pipelined = engine.prefers_lookahead()
    if max_iter < 1:
        return h

    def record(i, dJ, dPops):
        h.dJ.append(dJ)
        h.dPops.append(dPops if i > n_lambda_only else float('nan'))
        if log:
            log('Iteration %.3d: dJ: %.2e, dPops: %s' % (i, dJ, 'Just iterating Jbar' if i <= n_lambda_only else '%.2e' % dPops))
        if not (np.isfinite(dJ) and np.isfinite(dPops)):
            h.nonfinite = True

    dJ, dPops, i = 1.0, 1.0, 0
    if not pipelined:
        while dJ > dJ_tol or dPops > dPops_tol:
            i += 1
            dJ, dP = mali_step(engine, i > n_lambda_only, reducer)
            if dP is not None:
                dPops = dP
            record(i, dJ, dPops)
            if i >= max_iter:
                break
        h.converged = (dJ <= dJ_tol and dPops <= dPops_tol)
        return h

    i = 1
    engine.formal_sol_gamma_async()
    if i > n_lambda_only:
        engine.stat_equil_async()
    _begin(engine, reducer)
    can_speculate = True
    while True:
        speculating = can_speculate and i < max_iter
        if speculating:
            try:
                engine.formal_sol_gamma_speculative()          # iteration i + 1, ahead of the decision
            except Exception as e:
                from ._capi import LSX_EUNSUPPORTED
                if getattr(e, 'code', None) != LSX_EUNSUPPORTED:
                    # a failed enqueue: collect the read-back in flight, take back whatever part of the call was enqueued
                    # (the library has already switched buffers), and let the caller see the error
                    try:
                        _end(engine, reducer)
                    finally:
                        try:
                            engine.discard_formal_sol()
                        except Exception:
                            pass
                    raise
                # LSX_EUNSUPPORTED is the library's refusal BEFORE anything is enqueued (frozen columns): the loop goes on
                # without looking ahead
                can_speculate = speculating = False
        try:
            dJ, dP = _end(engine, reducer)                      # monitors of iteration i
        except Exception:
            if speculating:
                engine.discard_formal_sol()
            raise
        if i > n_lambda_only:
            dPops = dP
        record(i, dJ, dPops)
        if not (dJ > dJ_tol or dPops > dPops_tol) or i >= max_iter:
            if speculating:
                engine.discard_formal_sol()                     # the loop ends with iteration i: its I, J, Gamma are the results
            break
        i += 1
        if not speculating:
            engine.formal_sol_gamma_async()
        if i > n_lambda_only:
            engine.stat_equil_async()
        _begin(engine, reducer)
    h.converged = (dJ <= dJ_tol and dPops <= dPops_tol)
    return h


def response_function(I_plus, I_minus, I_base, mu_index=-1):
    """response_fn.py:59-67: rf[la, k] = (I+[la, mu] - I-[la, mu]) / I_base[la, mu].
    I_plus / I_minus: [Nspace][Nspect][Nrays] (one converged run per perturbed depth),
    I_base: [Nspect][Nrays]."""
    Ip = np.asarray(I_plus)[:, :, mu_index].T      # [Nspect][Nspace]
    Im = np.asarray(I_minus)[:, :, mu_index].T
    return (Ip - Im) / np.asarray(I_base)[:, mu_index][:, None]


def iterate_mali_columns(engine, dJ_tol=2e-3, dPops_tol=1e-3, n_lambda_only=3, max_iter=500, all_done=None, log=None,
                         report=None):
    """Many independent columns, each with the reference's own stopping rule: a column is frozen
    (lsx_set_active_columns) as soon as ITS dJ <= 2e-3 and dPops <= 1e-3, so it performs exactly the
    iterations a single-column reference Context would (test.py:20-29, response_fn.py:11-21).

    engine: problem.Engine.  all_done(bool) -> bool: hook for the multi-GPU driver (logical AND over
    ranks).  Returns the number of iterations each column took ([ncol] int array).

    A column whose monitors turn NaN is treated as the reference's `while dJ > 2e-3 or dPops > 1e-3` treats it: the
    comparisons with NaN are False, so the column stops as soon as its other monitor is below its threshold (it does
    not keep the batch iterating to max_iter).  report (a dict, optional) receives 'nonfinite': the indices of the
    columns that ever showed a non-finite monitor."""
    from . import _capi
    ncol = engine.ncol
    active = np.ones(ncol, dtype=bool)
    dP = np.ones(ncol)
    n_iter = np.zeros(ncol, dtype=np.int64)
    bad = np.zeros(ncol, dtype=bool)
    i = 0
    engine.set_active_columns(None)
    while True:
        i += 1
        engine.formal_sol_gamma()
        dJ = engine.get(_capi.LSX_DJ_COL)
        if i > n_lambda_only:
            engine.stat_equil()
            dP = np.where(active, engine.get(_capi.LSX_DPOPS_COL), dP)
        n_iter[active] = i
        bad |= active & ~(np.isfinite(dJ) & np.isfinite(dP))
        with np.errstate(invalid='ignore'):
            still = (dJ > dJ_tol) | (dP > dPops_tol)      # NaN compares False (test.py:23)
        active &= still
        if i >= max_iter:
            active[:] = False
        done = not active.any()
        if all_done is not None:
            done = all_done(done)
        if log:
            log('Iteration %.3d: %d of %d columns still iterating' % (i, int(active.sum()), ncol))
        if done:
            break
        engine.set_active_columns(active)
    engine.set_active_columns(None)
    if report is not None:
        report['nonfinite'] = np.flatnonzero(bad)
    return n_iter
