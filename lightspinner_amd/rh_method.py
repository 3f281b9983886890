"""Drop-in counterpart of Lightspinner's rh_method.Context (rh_method.py:490-745).

    ctx = Context(atmos, spect, eqPops, background)
    dJ = ctx.formal_sol_gamma_matrices()
    dPops = ctx.stat_equil()
    ctx.I, ctx.J, ctx.activeAtoms[0].n / .Gamma / .C / .nStar / .nTotal / .trans[kr].phi ...

The constructor reads the same (duck-typed) attributes the reference's constructor reads from
Lightspinner's Atmosphere / SpectrumConfiguration / AtomicStateTable / Background objects.  Nothing
numerical happens on the host: everything between the atmosphere and the hot path's inputs is
evaluated by the library,

  * setup='native' (chosen by 'auto' when the model objects carry their atomic data: levels with
    E_SI / g / stage, lines with gRad / stark / vdw, collisions that are Omega / CI / CE tables):
    the data go down once (lsx_set_atomic_data) and lsx_set_atmosphere derives vBroad, aDamp, the
    line profiles and the collisional rates on the device (rh_method.py:198-243, 474-487;
    atomic_model.py:66-69, 491-502);
  * setup='methods' (models that only offer the reference's methods): `atom.v_broad`,
    `line.damping` and `collision.compute_rates` are CALLED, as the reference calls them, and
    their results handed over; the profiles are still built on the device
    (lsx_set_line_profiles).

`t.phi`, `t.wphi`, `t.aDamp`, `atom.vBroad` and `atom.C` are read back from the library.  The two
hot calls run on the GPU through the lsx C ABI; there is no CPU fallback (the HIP library must be
built).

Read-back (round 6).  The reference's `ctx.J`, `ctx.I` and `atom.Gamma` are live arrays that every call rewrites in
place; its drivers (test.py:20-29, response_fn.py:11-21) read them once, after the loop.  Here they are fetched from the
device when they are first LOOKED AT: until then a call costs the enqueue, the wait and one 16-byte monitor read-back.
An array that has been handed out once is kept current after every call from then on (its holder may read it, or write it,
at any time: edits are found by comparing with what was last synchronised, as before) -- `readback='eager'` asks for that
from the start.  `atom.n` is the caller's own array (`eqPops[name].pops`, rh_method.py:412-416): stat_equil writes it
in place after every call, always.

Look-ahead (round 6).  test.py:20-29 calls the two verbs in turn; each returns a number the loop's condition needs, so each is a
host round trip with the GPU idle behind it.  stat_equil() here enqueues the NEXT formal solution speculatively behind its own
read-back (lsx_sync_begin_populations / lsx_formal_sol_gamma_speculative, include/lsx.h) before it returns: if the caller's next
call is formal_sol_gamma_matrices() -- and it has not edited n or J in between -- that call only waits for it.  Anything else
(another stat_equil, a look at J / I / Gamma, an edit, update_collisions) takes the speculative call back first
(lsx_discard_formal_sol): every result is what the plain sequence gives, bit for bit (tests/context_cases.py).  `lookahead=False`
switches it off.

Deliberate differences, all outside the numbers the drivers use:
  * `t.Rij` / `t.Rji` are not produced (the reference accumulates them without ever zeroing or
    reading them, rh_method.py:691-692); accessing them raises AttributeError.
  * collisional rates are evaluated when the Context is built and whenever
    `update_collisions()` is called, not on every formal solution (they depend on the
    atmosphere only, rh_method.py:474-487).
"""
from typing import List, Optional

import numpy as np

from . import _capi, atomdata
from . import constants as Const
from .problem import Problem, Transition, ColumnBlock, Engine

_KNOWN_COLLISIONS = ('Omega', 'CI', 'CE')


def _same(a, b):
    if a is b:
        return True
    try:
        return bool(a == b)
    except Exception:
        return False


def _contains(seq, item):
    return any(_same(x, item) for x in seq)


def _is_line(trans):
    return hasattr(trans, 'Aji') and hasattr(trans, 'lambda0')


def models_carry_atomic_data(models, atmos) -> bool:
    """can lsx_set_atomic_data be filled from these objects (atomdata.from_models)?"""
    if not all(hasattr(atmos, k) for k in ('ne', 'vturb')):
        return False
    for m in models:
        table = getattr(m, 'atomicTable', None)
        if table is None:
            return False
        try:
            table[m.name].weight, table['H'].weight, table['He'].abundance
        except Exception:
            return False
        if not all(hasattr(l, 'E_SI') and hasattr(l, 'g') and hasattr(l, 'stage') for l in m.levels):
            return False
        if not all(hasattr(l, 'gRad') and hasattr(l, 'stark') for l in m.lines):
            return False
        if not all(type(c).__name__ in _KNOWN_COLLISIONS and hasattr(c, 'temperature') and hasattr(c, 'rates')
                   for c in m.collisions):
            return False
    return True


class ComputationalTransition:
    """rh_method.py:25-288 (state only; uv() lives in the sweep kernel, compute_phi in the set-up kernels)."""

    def __init__(self, trans, compAtom: 'ComputationalAtom', atmos, spect):
        self.transModel = trans
        self.atom = compAtom
        self.wavelength = np.asarray(trans.wavelength, dtype=np.float64)
        self.isLine = _is_line(trans)
        if self.isLine:
            self.Aji, self.Bji, self.Bij, self.lambda0 = float(trans.Aji), float(trans.Bji), float(trans.Bij), float(trans.lambda0)
        else:
            self.alpha = np.asarray(trans.alpha, dtype=np.float64)
        self.i, self.j = int(trans.i), int(trans.j)
        self.Nblue = int(np.searchsorted(spect.wavelength, self.wavelength[0]))       # :122
        self.active = np.zeros(spect.wavelength.shape[0], dtype=bool)                  # :124-127
        for la, s in enumerate(spect.activeSet):
            if _contains(s, trans):
                self.active[la] = True
        self.gij = None
        self._line_index = None          # position among the lines of the transition table
        self._phi_offset = None          # first row of this line in LSX_PHI

    def lt(self, la: int) -> int:
        return la - self.Nblue

    def wlambda(self, la: Optional[int] = None):
        """rh_method.py:157-196: trapezoid weights of the transition's own grid, in Doppler units for a line"""
        lam = self.wavelength
        w = np.empty_like(lam)
        w[0], w[-1] = 0.5 * (lam[1] - lam[0]), 0.5 * (lam[-1] - lam[-2])
        w[1:-1] = 0.5 * (lam[2:] - lam[:-2])
        if self.isLine:
            w *= Const.CLight / self.lambda0
        return w if la is None else w[la]

    # -- read back from the library ------------------------------------------------------
    def _need_line(self):
        if not self.isLine:
            raise AttributeError('a continuum has no line profile')
        return self.atom._context

    @property
    def phi(self):
        """[Nlambda][Nrays][2][Nspace] (rh_method.py:224)"""
        ctx = self._need_line()
        p = ctx.problem
        phi = ctx._profiles()[0][self._phi_offset:self._phi_offset + self.wavelength.shape[0]]
        if p.phi_compact:
            phi = np.broadcast_to(phi[:, None, None, :], (phi.shape[0], p.Nrays, 2, p.Nspace))
        return phi

    @property
    def wphi(self):
        return self._need_line()._profiles()[1][self._line_index]

    @property
    def aDamp(self):
        return self._need_line()._damping()[self._line_index]


class ComputationalAtom:
    """rh_method.py:290-487."""

    def __init__(self, atom, atmos, spect, eqPops):
        self.atomicModel = atom
        self.atomicTable = getattr(eqPops, 'atomicTable', None)
        self.spect = spect
        self.atmos = atmos
        self.pops = eqPops[atom.name]
        self.hPops = eqPops['H']
        self.nTotal = self.pops.nTotal
        self.trans: List[ComputationalTransition] = []
        for l in atom.lines:                                   # lines first, then continua, :399-405
            if _contains(spect.transitions, l):
                self.trans.append(ComputationalTransition(l, self, atmos, spect))
        for c in atom.continua:
            if _contains(spect.transitions, c):
                self.trans.append(ComputationalTransition(c, self, atmos, spect))
        Nlevel = len(atom.levels)
        self.Nlevel = Nlevel
        self.Ntrans = len(self.trans)
        self._Gamma = np.zeros((Nlevel, Nlevel, atmos.Nspace))
        self.nStar = self.pops.nStar
        if self.pops.pops is not None:                         # warm start, :412-416
            self.n = self.pops.pops
        else:
            self.n = np.copy(self.nStar)
            self.pops.pops = self.n
        self._context = None
        self._index = None

    @property
    def Gamma(self):
        """[Nlevel][Nlevel][Nspace], rh_method.py:587-590, 698-703 (fetched when first looked at, kept current from then on)"""
        ctx = self._context
        if ctx is None:
            return self._Gamma
        return ctx._lazy_out('Gamma')[self._index]

    @property
    def vBroad(self):
        return self._context._broadening()[self._index]

    @property
    def C(self):
        """[Nlevel][Nlevel][Nspace] collisional rates as the library holds them (rh_method.py:474-487)"""
        ctx = self._context
        return ctx._engine.gamma_of_atom(ctx._engine.get(_capi.LSX_C), self._index)[0]


class Context:
    """rh_method.py:490-745 on the GPU."""

    def __init__(self, atmos, spect, eqPops, background, device: int = 0, stream=None, lib=None, setup: str = 'auto',
                 formal_solver: str = 'linear', readback: str = 'lazy', lookahead: Optional[bool] = None):
        self.atmos = atmos
        self.atmos.nondimensionalise()
        self.spect = spect
        self.background = background
        self.eqPops = eqPops
        self.activeAtoms: List[ComputationalAtom] = [ComputationalAtom(a, atmos, spect, eqPops)
                                                     for a in spect.radSet.activeAtoms]
        Nspect, Nspace, Nrays = spect.wavelength.shape[0], atmos.Nspace, atmos.Nrays
        if readback not in ('lazy', 'eager'):
            raise ValueError("readback must be 'lazy' or 'eager'")
        # the result arrays (one object each for the life of the context, rewritten in place like the reference's), whether the
        # device holds something newer (`_stale`), and whether somebody outside holds them (`_handed`: kept current and checked for edits)
        self._host = {'J': np.zeros((Nspect, Nspace)), 'I': np.zeros((Nspect, Nrays)),
                      'Gamma': [a._Gamma for a in self.activeAtoms]}
        self._stale = {'J': False, 'I': False, 'Gamma': False}
        self._handed = {k: readback == 'eager' for k in self._host}
        self._J_synced = None
        models = [a.atomicModel for a in self.activeAtoms]
        if setup == 'auto':
            setup = 'native' if models_carry_atomic_data(models, atmos) else 'methods'
        if setup not in ('native', 'methods'):
            raise ValueError("setup must be 'auto', 'native' or 'methods'")
        self.setup = setup

        # ---- flatten to the lsx problem description --------------------------------------
        trans, active = [], []
        nline = off = 0
        for a, atom in enumerate(self.activeAtoms):
            atom._context, atom._index = self, a
            for t in atom.trans:
                tr = Transition(atom=a, is_line=t.isLine, i=t.i, j=t.j, Nblue=t.Nblue, Nlambda=t.wavelength.shape[0])
                if t.isLine:
                    tr.Aji, tr.Bji, tr.Bij, tr.lambda0 = t.Aji, t.Bji, t.Bij, t.lambda0
                    t._line_index, t._phi_offset = nline, off
                    nline += 1
                    off += t.wavelength.shape[0]
                else:
                    tr.alpha = t.alpha
                trans.append(tr)
                active.append(t.active)
        sca = np.asarray(background.sca, dtype=np.float64)
        sca_per_lambda = not (sca.ndim == 2 and np.all(sca == sca[0:1]))
        phi_compact = bool(np.all(np.asarray(atmos.vlos) == 0.0))
        self.problem = Problem(Nspace=Nspace, wavelength=np.asarray(spect.wavelength), muz=atmos.muz, wmu=atmos.wmu,
                               Nlevel=[a.Nlevel for a in self.activeAtoms], trans=trans,
                               active=np.array(active, dtype=bool).reshape(len(trans), Nspect),
                               sca_per_lambda=sca_per_lambda, phi_compact=phi_compact,
                               atom_names=[a.atomicModel.name for a in self.activeAtoms])
        self._engine = Engine(self.problem, 1, device=device, stream=stream, lib=lib)
        if formal_solver != 'linear':          # 'parabolic': the monotonic piecewise-parabolic rule (include/lsx.h, N4; not in the reference)
            self._engine.set_formal_solver(formal_solver)
        self._cache = {}
        if setup == 'native':
            in_table = lambda m, l: _contains(spect.transitions, l)
            self._engine.set_atomic_data(atomdata.from_models(models, line_filter=in_table))
        self._upload()
        # stat_equil enqueues the next formal solution ahead where the library says that pays (one launch chain on one stream)
        self._lookahead = self._engine.prefers_lookahead() if lookahead is None else bool(lookahead)
        self._spec = False               # a speculative formal solution is enqueued and has not been accepted or taken back
        self._n_synced = [np.array(a.n, dtype=np.float64) for a in self.activeAtoms]
        if self._handed['J']:
            self._J_synced = self._host['J'].copy()

    # -- the live result arrays ------------------------------------------------------------
    def _cancel_lookahead(self):
        """take the speculative formal solution back: the library's results are the last ACCEPTED call's again"""
        if self._spec:
            self._engine.discard_formal_sol()
            self._spec = False

    def _fetch(self, what):
        self._cancel_lookahead()
        eng = self._engine
        if what == 'J':
            self._host['J'][...] = eng.get(_capi.LSX_J)[0]
            if self._handed['J']:
                self._J_synced = self._host['J'].copy()
        elif what == 'I':
            self._host['I'][...] = eng.get(_capi.LSX_I)[0]
        else:
            G = eng.get(_capi.LSX_GAMMA)
            for a, arr in enumerate(self._host['Gamma']):
                arr[...] = eng.gamma_of_atom(G, a)[0]
        self._stale[what] = False

    def _lazy_out(self, what):
        """the array, current; from now on somebody outside may hold it"""
        if self._stale[what]:
            self._fetch(what)
        if not self._handed[what]:
            self._handed[what] = True
            if what == 'J':
                self._J_synced = self._host['J'].copy()
        return self._host[what]

    @property
    def J(self):
        """[Nspect][Nspace] mean intensity, rh_method.py:562, 640"""
        return self._lazy_out('J')

    @J.setter
    def J(self, value):
        self._lazy_out('J')[...] = value          # found by the next call's comparison and sent down

    @property
    def I(self):
        """[Nspect][Nrays] emergent intensity, rh_method.py:563, 638"""
        return self._lazy_out('I')

    def _results_changed(self, names):
        """a call has rewritten these on the device: refresh the ones somebody holds, mark the others"""
        for k in names:
            if self._handed[k]:
                self._fetch(k)
            else:
                self._stale[k] = True

    # -- packing -------------------------------------------------------------------------
    def _cat_n(self):
        return np.concatenate([np.asarray(a.n, dtype=np.float64) for a in self.activeAtoms], axis=0)

    def _row(self, x):
        return np.asarray(x, dtype=np.float64)[None]

    def _upload(self):
        """atmosphere, populations, background -> library; then the set-up chain (native) or the models' own numbers
        (methods) -> vBroad, aDamp, profiles, collisional rates"""
        p, atmos, bg = self.problem, self.atmos, self.background
        sca = np.asarray(bg.sca, dtype=np.float64)
        sca = sca if p.sca_per_lambda else (sca[0] if sca.ndim == 2 else sca)
        native = self.setup == 'native'
        if native:
            C = np.zeros((p.NL2tot, p.Nspace))
        else:
            C = np.concatenate([self._model_collisions(a).reshape(-1, p.Nspace) for a in self.activeAtoms], axis=0)
        nTotal = np.stack([np.asarray(a.nTotal, dtype=np.float64) for a in self.activeAtoms])
        block = ColumnBlock(
            height=self._row(atmos.height), temperature=self._row(atmos.temperature),
            nStar=np.concatenate([np.asarray(a.nStar) for a in self.activeAtoms], axis=0)[None], nTotal=nTotal[None],
            n=self._cat_n()[None], C=C[None], bg_chi=self._row(bg.chi), bg_eta=self._row(bg.eta), bg_sca=sca[None],
            phi=None, wphi=None)
        self._engine.set_columns(0, block)
        vlos = None if p.phi_compact else self._row(atmos.vlos)
        self._cache = {}
        if native:
            hGround = self.activeAtoms[0].hPops.n[0] if self.activeAtoms else np.zeros(p.Nspace)
            self._engine.set_atmosphere(0, self._row(atmos.temperature), self._row(atmos.ne), self._row(atmos.vturb),
                                        self._row(hGround), nTotal[None], vlos=vlos, lte_pops=False)
        elif p.Nlines:
            vB = np.stack([np.asarray(a.atomicModel.v_broad(atmos), dtype=np.float64) for a in self.activeAtoms])
            aD = np.stack([np.asarray(t.transModel.damping(atmos, vB[a], atom.hPops.n[0])[0], dtype=np.float64)
                           for a, atom in enumerate(self.activeAtoms) for t in atom.trans if t.isLine])
            self._engine.set_line_profiles(0, aD[None], vB[None], vlos)
            self._cache.update(vBroad=vB, aDamp=aD)

    def _model_collisions(self, atom):
        """rh_method.py:474-487 through the models' own compute_rates"""
        C = np.zeros((atom.Nlevel, atom.Nlevel, self.problem.Nspace))
        for col in atom.atomicModel.collisions:
            col.compute_rates(self.atmos, atom.nStar, C)
        C[C < 0.0] = 0.0
        return C

    # -- read-back of what the library derived ---------------------------------------------
    def _profiles(self):
        if 'phi' not in self._cache:
            self._cache['phi'] = (self._engine.get(_capi.LSX_PHI)[0], self._engine.get(_capi.LSX_WPHI)[0])
        return self._cache['phi']

    def _damping(self):
        if 'aDamp' not in self._cache:
            self._cache['aDamp'] = self._engine.get(_capi.LSX_ADAMP)[0]
        return self._cache['aDamp']

    def _broadening(self):
        if 'vBroad' not in self._cache:
            self._cache['vBroad'] = self._engine.get(_capi.LSX_VBROAD)[0]
        return self._cache['vBroad']

    def update_collisions(self):
        """re-derive everything that depends on the atmosphere alone (the reference recomputes the collisional rates
        on every formal solution, rh_method.py:589) and keep J"""
        self._cancel_lookahead()
        if self._stale['J']:
            self._fetch('J')
        J = self._host['J'].copy()
        self._upload()
        self._engine.set(_capi.LSX_J, J[None])

    def _host_edits(self):
        """the reference's arrays are live numpy objects the caller may edit between calls.  The populations are the caller's own
        arrays (eqPops[...].pops): compared on every call (a few kB).  J can only have been edited if it has been handed out.
        -> (n edited, J edited)"""
        n_ed = any(not np.array_equal(a.n, s) for a, s in zip(self.activeAtoms, self._n_synced))
        J_ed = self._handed['J'] and not self._stale['J'] and not np.array_equal(self._host['J'], self._J_synced)
        return n_ed, J_ed

    def _push_host_edits(self, edits=None):
        n_ed, J_ed = self._host_edits() if edits is None else edits
        if n_ed:
            self._engine.set(_capi.LSX_N, self._cat_n()[None])
            self._n_synced = [np.array(a.n, dtype=np.float64) for a in self.activeAtoms]
        if J_ed:
            self._engine.set(_capi.LSX_J, self._host['J'][None])
            self._J_synced = self._host['J'].copy()

    # -- the two verbs ---------------------------------------------------------------------
    def formal_sol_gamma_matrices(self) -> float:
        """rh_method.py:565-708 -> max relative change of J"""
        edits = self._host_edits()
        if self._spec and not any(edits):
            # the formal solution stat_equil enqueued ahead IS this call: wait for it
            self._spec = False
            dJ = self._engine.sync()[0]
        else:
            self._cancel_lookahead()
            self._push_host_edits(edits)
            dJ = self._engine.formal_sol_gamma()
        self._results_changed(('J', 'I', 'Gamma'))
        return dJ

    def stat_equil(self) -> float:
        """rh_method.py:710-745 -> max relative population change; populations are written back IN
        PLACE into the arrays that alias eqPops[...].pops (rh_method.py:412-416, response_fn.py:62)"""
        self._cancel_lookahead()                  # (a second stat_equil in a row works on the same Gamma, as in the reference)
        self._push_host_edits()
        eng = self._engine
        if self._lookahead:
            eng.stat_equil_async()
            eng.sync_begin(populations=True)      # monitors + n, read back behind the solve ...
            try:
                eng.formal_sol_gamma_speculative()    # ... and the next iteration's formal solution behind the read-back
                self._spec = True
            except _capi.LsxError:
                self._spec = False
            try:
                dPops = eng.sync_end()[1]
            except _capi.LsxError:
                self._cancel_lookahead()          # (a singular system, rh_method.py:739: nothing is built on it)
                raise
            n = eng.fetch_populations()[0]
        else:
            dPops = eng.stat_equil()
            n = eng.get(_capi.LSX_N)[0]
        off = 0
        for a, atom in enumerate(self.activeAtoms):
            atom.n[...] = n[off:off + atom.Nlevel]
            self._n_synced[a] = n[off:off + atom.Nlevel]
            off += atom.Nlevel
        return dPops

    def close(self):
        self._spec = False
        self._engine.close()
