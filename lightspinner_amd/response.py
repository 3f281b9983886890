"""Brute-force NLTE temperature response function (response_fn.py:23-74) as ONE batch of columns:
the base column plus, for every perturbed depth k, the atmosphere with T[k] +/- dT/2, all warm
started from the converged base populations (response_fn.py:33).  The 2 x Nspace MALI solves the
reference runs one after the other are independent columns here.

Set-up quantities of a temperature perturbation at depth k differ from the base column at depth k
only (every set-up formula is pointwise in depth and the uniform height shift of
atmosphere.py:104-105 cancels in |z_k - z_k+1|), so a perturbed column is the base
column plus a "delta": the depth-k entries of the arrays that change."""
import numpy as np

from .problem import ColumnBlock, Problem

# names of the per-column arrays a delta may touch and where they live in a ColumnBlock
_PER_ATOM = {'nStar': 'nStar', 'C': 'C', 'nTotal': 'nTotal'}


_DELTA_FIELDS = ('height', 'temperature', 'nStar', 'nTotal', 'n', 'C', 'bg_chi', 'bg_eta', 'bg_sca', 'phi', 'wphi')


def _line_layout(prob: Problem):
    """-> ({transition: line index}, {transition: first row of its block in phi})"""
    line_idx, phi_off = {}, {}
    o = li = 0
    for kr, t in enumerate(prob.trans):
        if t.is_line:
            line_idx[kr], phi_off[kr] = li, o
            o += t.Nlambda
            li += 1
    return line_idx, phi_off


def _apply_into(prob: Problem, out: dict, c: int, delta: dict, k: int, layout) -> None:
    """writes the depth-k entries of `delta` into column c of the arrays of `out` (ColumnBlock fields)"""
    lev_off, lev2_off = prob.lev_off, prob.lev2_off
    line_idx, phi_off = layout
    for key, v in delta.items():
        v = np.asarray(v)
        if key == 'temperature':
            out['temperature'][c, k] = v
        elif key in ('bg_chi', 'bg_eta'):
            out[key][c, :, k] = v
        elif key == 'bg_sca':
            if prob.sca_per_lambda:
                out['bg_sca'][c, :, k] = v
            else:
                out['bg_sca'][c, k] = v
        elif key[0] == 'a' and '_' in key:
            a, name = key[1:].split('_', 1)
            a = int(a)
            nl = prob.Nlevel[a]
            if name == 'nStar':
                out['nStar'][c, lev_off[a]:lev_off[a] + nl, k] = v
            elif name == 'nTotal':
                out['nTotal'][c, a, k] = v
            elif name == 'C':
                out['C'][c, lev2_off[a]:lev2_off[a] + nl * nl, k] = v.reshape(-1)
            # vBroad / weight only matter through phi, which the delta carries explicitly
        elif key[0] == 't' and '_' in key:
            kr, name = key[1:].split('_', 1)
            kr = int(kr)
            if name == 'wphi':
                out['wphi'][c, line_idx[kr], k] = v
            elif name == 'phi':
                t = prob.trans[kr]
                sl = slice(phi_off[kr], phi_off[kr] + t.Nlambda)
                if prob.phi_compact:
                    out['phi'][c, sl, k] = v
                else:
                    out['phi'][c, sl, :, :, k] = v if v.ndim == 3 else v[:, None, None]


def apply_delta(prob: Problem, base: ColumnBlock, delta: dict, k: int, start_n=None) -> ColumnBlock:
    """base: ColumnBlock with ncol == 1.  delta: {array name as in the problem-file format
    (fixtures.py): values at depth k}.  Returns the perturbed column."""
    return apply_deltas(prob, base, [(delta, k)], start_n=start_n)


def apply_deltas(prob: Problem, base: ColumnBlock, deltas, start_n=None) -> ColumnBlock:
    """the perturbed columns of a list of (delta, depth k) pairs as ONE block: the base column repeated, then every delta's
    depth-k entries written in place (one copy of the base arrays for the whole batch: building the 164 columns of the CaII
    response function one ColumnBlock at a time and concatenating them was a quarter of its wall time)"""
    n = len(deltas)
    out = {f: np.repeat(np.asarray(getattr(base, f))[:1], n, axis=0) for f in _DELTA_FIELDS}
    layout = _line_layout(prob)
    for c, (delta, k) in enumerate(deltas):
        _apply_into(prob, out, c, delta, k, layout)
    if start_n is not None:
        out['n'][:] = start_n
    return ColumnBlock(**out).validate(prob)


_RESULT_KEYS = ('I', 'n', 'niter', 'traj_dJ', 'traj_dPops')


def deltas_of(fixture: dict, k: int, tag: str) -> dict:
    """the delta of one perturbed run ('p': T[k] + dT/2, 'm': T[k] - dT/2) out of a fixture written by
    tests/golden/make_golden.py (gen_rf / gen_rf_inputs)"""
    pre = 'k%d%s_' % (k, tag)
    return {key[len(pre):]: v for key, v in fixture.items() if key.startswith(pre) and key[len(pre):] not in _RESULT_KEYS}


def index_deltas(fixture: dict) -> dict:
    """{(k, tag): delta} of every perturbed run of a fixture in ONE pass over its keys (deltas_of scans all of them per run: for the
    164 runs of the CaII response function that was 0.6 million string comparisons, two thirds of the function's wall time)"""
    import re
    pat = re.compile(r'k(\d+)([pm])_(.+)$')
    out = {}
    for key, v in fixture.items():
        m = pat.match(key)
        if m and m.group(3) not in _RESULT_KEYS:
            out.setdefault((int(m.group(1)), m.group(2)), {})[m.group(3)] = v
    return out


def run_response_function(prob: Problem, base: ColumnBlock, fixture: dict, ks, lib=None, device=0, mu_index=-1, log=None,
                          rank=0, world=1, all_done=None, gather=None, stream=None):
    """response_fn.py:11-74 as two batched solves: (1) the base column to convergence (test.py:20-29 loop, per-column
    stopping rule), (2) the 2 x len(ks) perturbed columns, warm started from the base populations
    (response_fn.py:33), every column with its own stopping rule, then rf[la, k] = (I+ - I-) / I_base at `mu_index`
    (response_fn.py:59-67).  -> dict(rf [Nspect][len(ks)], I_base, n_base, n_iter_base, n_iter [2 len(ks)], I [...]).

    Over several ranks (165 columns -> 21 / 20 per GPU): the perturbed columns are block partitioned
    (parallel.shard_columns); every rank solves the (tiny) base column itself -- bit-identical everywhere, so no
    broadcast of the warm-start populations is needed -- and iterates its shard until ALL ranks are done
    (all_done = parallel.AllDone(): logical AND, the per-column stopping rule needs no other exchange);
    gather(I_local [n_local][Nspect][Nrays], n_iter_local) -> (I of all columns, n_iter of all columns) on every rank
    (e.g. torch.distributed.all_gather_object).  A rank with an empty shard still takes part in the collectives."""
    from . import _capi, drivers
    from .parallel import shard_columns
    from .problem import Engine
    e0 = Engine(prob, 1, device=device, lib=lib, stream=stream)
    e0.set_columns(0, base.slice(0, 1))
    # (one column: its own stopping rule IS the global one -- the engine loop, which keeps the next formal solution enqueued while the
    # host reads the monitors where the library says that pays, lsx_prefers_lookahead; the same iterations and bits)
    it0 = [drivers.iterate_mali_engine(e0, log=log).n_iter]
    I_base, n_base = e0.get(_capi.LSX_I)[0], e0.get(_capi.LSX_N)[0]
    jobs = [(int(k), tag) for k in ks for tag in ('p', 'm')]
    if world > 1:
        # start-up check of the sharded job: every rank's engines must be made alike (options, rule, plan, and the mapping the
        # whole problem's column count selects) -- on the base engine, which every rank has, even one with an empty shard
        from .parallel import check_same_options
        e0.set_sweep_policy('auto', len(jobs))
        check_same_options(e0)
    e0.close()
    first, count = shard_columns(len(jobs), rank, world)
    mine = jobs[first:first + count]
    if mine:
        index = index_deltas(fixture)
        batch = apply_deltas(prob, base, [(index.get((k, tag), {}), k) for k, tag in mine], start_n=n_base)
        # the kernel choice belongs to the problem -- all 2 len(ks) perturbed columns -- not to this rank's shard of it
        eng = Engine(prob, batch.ncol, device=device, lib=lib, stream=stream, policy_columns=len(jobs))
        for a in range(0, batch.ncol, 64):
            eng.set_columns(a, batch.slice(a, min(batch.ncol, a + 64)))
        n_iter = drivers.iterate_mali_columns(eng, log=log, all_done=all_done)
        I = eng.get(_capi.LSX_I)
        eng.close()
    else:                                   # more ranks than columns: keep the collectives matched
        n_iter = np.zeros(0, dtype=np.int64)
        I = np.zeros((0, prob.Nspect, prob.Nrays))
        if all_done is not None:
            while not all_done(True):
                pass
    if world > 1:
        if gather is None:
            raise ValueError('run_response_function over several ranks needs gather=')
        I, n_iter = gather(I, n_iter)
    rf = drivers.response_function(I[0::2], I[1::2], I_base, mu_index=mu_index)
    return dict(rf=rf, I_base=I_base, n_base=n_base, n_iter_base=int(it0[0]), n_iter=n_iter, I=I, shard=(first, count))


def gather_over_ranks(group=None):
    """gather= for run_response_function: concatenates every rank's (I, n_iter) in rank order on every rank.  The
    payload is small (165 x Nspect x Nrays doubles for C5 = 1.9 MB in total): an object all-gather is enough."""
    import torch.distributed as dist

    def gather(I, n_iter):
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, (I, n_iter), group=group)
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
    return gather
