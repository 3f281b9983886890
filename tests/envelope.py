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


# ---- round 6: the bars BEHIND a statistical equilibrium are computed too ----------------------------------
# After stat_equil two faithful implementations differ by more than after a formal solution alone, for two reasons, and both can be
# COMPUTED from the oracle instead of typed into the tests:
#   (1) what they differed by before goes through the solve and the next formal solutions: the +-1-ulp-exp runs of `oracle_runs` go
#       through the same calls, so their spread after any call is the envelope there.  Behind the solve it holds in the maximum
#       norm, not entry by entry (the hook moves every exponential the same way; LU rounding does not follow that pattern):
#       `spread(...)`, K_ENVELOPE x the largest relative spread of the quantity.
#   (2) the LU itself.  rh_method.py:736-741 solves Gamma' n = b per depth; LAPACK's dgesv (the reference, through scipy), the oracle's
#       and the HIP kernel's dgetf2-ordered elimination may round differently.  The populations' response to that is the system's
#       componentwise condition number, cond(k) = max_i (|A^-1| |A| |n|)_i / |n_i| (Skeel; A = Gamma' with the eliminated row
#       replaced by ones): u cond is what ONE rounding of the data does.  Measured on the reference's own atoms (the reference's golden
#       populations against the oracle's, first statistical equilibrium): CaII 9.5e-11 = 0.38 u cond, MgII 1.2e-9 = 0.37 u cond,
#       iron 1.07e-8 = 0.46 u cond (cond = 2.1e8), carbon 5.8e-14 (cond 1e4: the envelope term decides there).
#   (3) what went in.  The Gamma a solve sees may differ by the single-call bar `tol` already, and -- behind an earlier statistical
#       equilibrium -- by what the populations differed by then, delta_n: the new populations are ratios of rates (<= 2 x the rates'
#       relative change), the rates are linear in I and in ratios of populations (<= 2 x).  In a problem with several active atoms the
#       atoms talk to each other through the radiation field: carbon (cond 1e4) inherits what iron (cond 2e8) feeds into J.
#       An intensity that is TRANSMITTED through tau optical depths answers a relative change of the opacity with tau times that
#       change (I ~ e^-tau): deep in the Lyman continuum J is 1e-29 -- twenty orders below the surface value -- and moves by
#       tau delta_n = 55 x 5e-11 (measured: 2.9e-9 at 30 nm, depth 41 of a Ca+H column, where the oracle's exp spread is 6e-15).
#       tau is bounded by the attenuation itself: T = 1 + ln(max_k J / J) per (column, wavelength, depth).
#   populations after a statistical equilibrium:   bar_n = K_ENVELOPE spread_n + K_LU u cond + 2 tol + 4 delta_n   (per atom; K_LU = 3)
#   J of a formal solution behind one (entry by entry):  bar = tol + K_ENVELOPE spread + 2 delta_n T
#   I (emergent: formed at tau ~ 1) behind one:          bar = tol + K_ENVELOPE spread + 2 delta_n
#   Gamma behind one (both measures of gamma_err):       bar = single-call bar + K_ENVELOPE spread + 4 delta_n
# with tol the single-call bar of the problem, spread the largest relative +-1-ulp spread of the quantity, delta_n = the deviation of
# the two implementations' populations MEASURED after the preceding statistical equilibrium (asserted below bar_n there; the largest over
# the atoms).
K_LU = 3.0
U_ROUND = 2.0 ** -53


def lu_condition(prob, Gamma, n_old, n_new):
    """per atom: the largest componentwise condition number of the statistical-equilibrium systems of rh_method.py:725-741 over
    columns and depths.  Gamma [ncol][NL2tot][Ns] as the solve saw it, n_old [ncol][NLtot][Ns] (decides the eliminated row: the first
    maximum), n_new the solution."""
    out = []
    Gamma, n_old, n_new = (np.asarray(x, dtype=np.float64) for x in (Gamma, n_old, n_new))
    off = 0
    for a in range(prob.Natoms):
        nl, o = prob.Nlevel[a], prob.lev2_off[a]
        A = Gamma[:, o:o + nl * nl, :].reshape(Gamma.shape[0], nl, nl, prob.Nspace).transpose(0, 3, 1, 2).copy()     # [col][k][l][l']
        no = n_old[:, off:off + nl, :].transpose(0, 2, 1)                                                        # [col][k][l]
        x = np.abs(n_new[:, off:off + nl, :].transpose(0, 2, 1))
        ie = np.argmax(no, axis=-1)                                                                              # first maximum
        ci, ki = np.indices(ie.shape)
        A[ci, ki, ie, :] = 1.0
        with np.errstate(all='ignore'):
            Ai = np.linalg.inv(A)
            cw = np.einsum('ckij,ckj->cki', np.abs(Ai), np.einsum('ckij,ckj->cki', np.abs(A), x)) / x
        cw = cw[np.isfinite(cw)]
        out.append(float(cw.max()) if cw.size else 0.0)
        off += nl
    return out


def spread(runs, call, what, measure):
    """`measure(x(+1 ulp), x(-1 ulp))` of the snapshot `what` after `call` (a maximum-norm measure: conftest.relerr, gamma_err)"""
    return measure(runs[1][call][what], runs[-1][call][what])


class SequenceBars:
    """Three oracle runs (exp as it is, +1 ulp, -1 ulp) through `ncalls` formal solutions with a statistical equilibrium behind
    call index >= se_from, and the LU's conditioning at every one of those: the computed bars of the header above.
    make_engine() -> a loaded oracle Engine."""

    def __init__(self, oracle_lib, make_engine, prob, ncalls, se_from, tol=1e-12):
        """tol: the single-call bar of this problem (1e-12: SURVEY 8d; 3e-11 where a ray crosses an interval next to w2's Taylor switch)"""
        self.prob, self.se_from, self.tol = prob, se_from, tol
        self.runs, self.cond = {}, {}
        try:
            for ulp in (0, 1, -1):
                oracle_lib.dll.lsx_oracle_set_exp_ulp(int(ulp))
                e = make_engine()
                snaps = []
                for it in range(ncalls):
                    dJ = e.formal_sol_gamma()
                    s = {w: e.get(w) for w in (_capi.LSX_I, _capi.LSX_J, _capi.LSX_GAMMA)}
                    s['dJ'] = dJ
                    if ulp == 0:
                        s[_capi.LSX_DJ_COL] = e.get(_capi.LSX_DJ_COL)
                    if se_from is not None and it >= se_from:
                        n_old = e.get(_capi.LSX_N)
                        s['dP'] = e.stat_equil()
                        s[_capi.LSX_N] = e.get(_capi.LSX_N)
                        if ulp == 0:
                            self.cond[it] = lu_condition(prob, s[_capi.LSX_GAMMA], n_old, s[_capi.LSX_N])
                    snaps.append(s)
                e.close()
                self.runs[ulp] = snaps
        finally:
            oracle_lib.dll.lsx_oracle_set_exp_ulp(0)

    def subset(self, ncol):
        """the bars of the first `ncol` columns of the ensemble (columns are independent problems: the runs' snapshots are sliced; the
        LU's condition number stays the larger set's maximum; the scalar monitors dJ / dP are maxima over ALL columns and are dropped)"""
        import copy
        b = copy.copy(self)
        b.runs = {u: [{k: (v[:ncol] if isinstance(v, np.ndarray) else None) for k, v in s.items()} for s in snaps] for u, snaps in self.runs.items()}
        return b

    def oracle(self, call, what):
        return self.runs[0][call][what]

    def _rel(self, a, b):
        a, b = np.asarray(a), np.asarray(b)
        return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))

    def n_bar(self, call, delta_n_prev=0.0):
        """largest admissible relative deviation of the populations after the statistical equilibrium behind `call`, per atom;
        delta_n_prev: what the populations (of any atom) differed by after the PREVIOUS statistical equilibrium (0: there was none)"""
        bars, off = [], 0
        for a in range(self.prob.Natoms):
            nl = self.prob.Nlevel[a]
            sp = self._rel(self.runs[1][call][_capi.LSX_N][:, off:off + nl], self.runs[-1][call][_capi.LSX_N][:, off:off + nl])
            bars.append(K_ENVELOPE * sp + K_LU * U_ROUND * self.cond[call][a] + 2.0 * self.tol + 4.0 * delta_n_prev)
            off += nl
        return bars

    def n_dev(self, n, ref, call=None):
        """measured relative deviation per atom"""
        n, ref = np.asarray(n), np.asarray(ref)
        out, off = [], 0
        for a in range(self.prob.Natoms):
            nl = self.prob.Nlevel[a]
            out.append(self._rel(n[..., off:off + nl, :], ref[..., off:off + nl, :]))
            off += nl
        return out

    def check_n(self, n, ref, call, who='', delta_n_prev=0.0):
        dev, bar = self.n_dev(n, ref), self.n_bar(call, delta_n_prev)
        assert all(d <= b for d, b in zip(dev, bar)), ('populations after the statistical equilibrium behind call %d%s: deviation per atom %s '
                                                       'above the computed bars %s (u cond = %s)' % (call + 1, who, dev, bar, [U_ROUND * c for c in self.cond[call]]))
        return max(dev)

    def _spread(self, call, what, floor=1e-300):
        a, b = self.runs[1][call][what], self.runs[-1][call][what]
        return float(np.max(np.abs(a - b) / np.maximum(np.abs(self.runs[0][call][what]), floor)))

    def I_bar(self, call, delta_n):
        """emergent intensity of formal solution `call` (relative, maximum norm); delta_n: the populations' measured deviation going in
        (0 before the first statistical equilibrium)"""
        return self.tol + K_ENVELOPE * self._spread(call, _capi.LSX_I) + 2.0 * delta_n

    def J_excess(self, J, ref, call, delta_n):
        """mean intensity of formal solution `call`, entry by entry against tol + K spread + 2 delta_n T, T = 1 + ln(max_k J / J) the
        optical depth the entry's radiation has been transmitted through at most.  -> (largest deviation / bar, largest relative
        deviation, the bar where T = 1)"""
        J, ref = np.asarray(J), np.asarray(ref)
        J0 = np.abs(self.runs[0][call][_capi.LSX_J])
        with np.errstate(divide='ignore', invalid='ignore'):
            T = 1.0 + np.log(np.maximum(np.max(J0, axis=-1, keepdims=True) / np.maximum(J0, 1e-300), 1.0))
            flat = self.tol + K_ENVELOPE * self._spread(call, _capi.LSX_J)
            bar = flat + 2.0 * delta_n * T
            rel = np.abs(J - ref) / np.maximum(np.abs(ref), 1e-300)
            rel = np.where(np.abs(ref) > 0, rel, np.where(J == ref, 0.0, np.inf))
        return float(np.max(rel / bar)), float(np.max(rel)), flat + 2.0 * delta_n

    def check_J(self, J, ref, call, delta_n, who=''):
        r, rel, flat = self.J_excess(J, ref, call, delta_n)
        assert r <= 1.0, ('J of call %d%s: %.2f x the computed bar (largest relative deviation %.2e; bar where nothing is transmitted %.2e)'
                          % (call + 1, who, r, rel, flat))
        return rel

    def gamma_bar(self, call, delta_n, gamma_err):
        eo, ed = gamma_err(self.runs[1][call][_capi.LSX_GAMMA], self.runs[-1][call][_capi.LSX_GAMMA], self.prob)
        return 10.0 * self.tol + K_ENVELOPE * eo + 4.0 * delta_n, self.tol + K_ENVELOPE * ed + 4.0 * delta_n
