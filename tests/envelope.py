"""The tolerance story of DESIGN.md 2, demonstrated instead of asserted (round 5).

formal_solver.py:36-43 forms w1 = (1 - e) - dtau e with e = exp(-dtau): next to the Taylor switch (dtau = 5e-4) that cancels to
dtau^2 / 2 = 1.25e-7, so a ONE-ulp difference between two correct exponentials (numpy's SIMD exp in the reference, libm's in the
oracle, the table-driven one of the HIP kernels) is an absolute 1.1e-16 on w1 = up to 9e-10 relative on that interval's
contribution.  No implementation can be asked to agree with another one more closely than that on a ray that crosses such an
interval -- but it CAN be asked to stay inside what a one-ulp change of exp() does to the oracle itself.

`oracle_runs` runs the oracle three times through the same sequence of calls: as it is, with every exp(-dtau) moved up by one ulp
and with every one moved down (oracle/lsx_oracle.c, lsx_oracle_set_exp_ulp: a test-only hook).  `inside` then checks an
implementation entry by entry:  |x - x_oracle| <= base |x_oracle| + K |x_(+1) - x_(-1)|  with K = 3 (the two exponentials of a pair
of implementations are each within an ulp of the true value and err independently from interval to interval, while the hook moves
all of them the same way: K covers sums whose signed responses partly cancel in the hook's run) and `base` the rounding-level bar
of SURVEY 8d that holds where no such interval is crossed (1e-12 on I and J)."""
import numpy as np

from lightspinner_amd import _capi

K_ENVELOPE = 3.0


def oracle_runs(oracle_lib, make_engine, ncalls, se_from=None, what=(_capi.LSX_I, _capi.LSX_J, _capi.LSX_GAMMA)):
    """make_engine() -> a loaded oracle Engine.  -> {ulp: [per call {what: array}]} for ulp in (0, +1, -1); statistical equilibrium
    after call index >= se_from (None: never)"""
    out = {}
    try:
        for ulp in (0, 1, -1):
            oracle_lib.dll.lsx_oracle_set_exp_ulp(int(ulp))
            e = make_engine()
            snaps = []
            for it in range(ncalls):
                e.formal_sol_gamma()
                snaps.append({w: e.get(w) for w in what})
                if se_from is not None and it >= se_from:
                    e.stat_equil()
            e.close()
            out[ulp] = snaps
    finally:
        oracle_lib.dll.lsx_oracle_set_exp_ulp(0)
    return out


def envelope(runs, call, what):
    """|x(+1 ulp) - x(-1 ulp)| entry by entry"""
    return np.abs(runs[1][call][what] - runs[-1][call][what])


def excess(x, runs, call, what, base, K=K_ENVELOPE, scale=None):
    """how far `x` lies outside base |x0| + K envelope, as a multiple of that bound (<= 1: inside), and the largest relative
    deviation / the largest envelope (relative) for the record.  scale: what `base` multiplies (default |x0| entry by entry)"""
    x0 = runs[0][call][what]
    env = envelope(runs, call, what)
    ref = np.abs(x0) if scale is None else scale
    bound = base * ref + K * env
    dev = np.abs(np.asarray(x) - x0)
    with np.errstate(divide='ignore', invalid='ignore'):
        ratio = np.where(bound > 0, dev / bound, np.where(dev > 0, np.inf, 0.0))
        rel = np.where(np.abs(x0) > 0, dev / np.abs(x0), 0.0)
        renv = np.where(np.abs(x0) > 0, env / np.abs(x0), 0.0)
    return float(np.max(ratio)), float(np.max(rel)), float(np.max(renv))


def inside(x, runs, call, what, base, K=K_ENVELOPE, scale=None):
    r, rel, renv = excess(x, runs, call, what, base, K, scale)
    assert r <= 1.0, ('outside the one-ulp-exp envelope of the oracle: %.2f x the bound (largest deviation %.2e relative, largest '
                      'envelope %.2e relative, base %.0e, K %g)' % (r, rel, renv, base, K))
    return rel, renv


def first_call_inside(oracle_lib, prob, block, I, J, solver='linear', base=1e-11, threads=8):
    """the first formal solution of (prob, block): I and J of an implementation against the oracle's, inside the one-ulp-exp envelope
    entry by entry.  -> (largest relative deviation of I, largest relative envelope of I)"""
    from lightspinner_amd import Engine

    def make():
        e = Engine(prob, block.ncol, lib=oracle_lib)
        e.set_columns(0, block)
        e.set_formal_solver(solver)
        oracle_lib.dll.lsx_oracle_set_threads(e._h, int(threads))
        return e
    runs = oracle_runs(oracle_lib, make, 1, what=(_capi.LSX_I, _capi.LSX_J))
    inside(J, runs, 0, _capi.LSX_J, base)
    return inside(I, runs, 0, _capi.LSX_I, base)
