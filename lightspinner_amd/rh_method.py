"""Drop-in counterpart of Lightspinner's rh_method.Context (rh_method.py:490-745).

    ctx = Context(atmos, spect, eqPops, background)
    dJ = ctx.formal_sol_gamma_matrices()
    dPops = ctx.stat_equil()
    ctx.I, ctx.J, ctx.activeAtoms[0].n / .Gamma / .C / .nStar / .nTotal / .trans[kr].phi ...

The constructor reads the same (duck-typed) attributes the reference's constructor reads from
Lightspinner's Atmosphere / SpectrumConfiguration / AtomicStateTable / Background objects and
calls the same model methods (`v_broad`, `damping`, `compute_rates`) -- the set-up maths of
ComputationalTransition.compute_phi and ComputationalAtom.compute_collisions is restated in
lineprofile.py / here.  The two hot calls run on the GPU through the lsx C ABI; there is no
CPU fallback (the HIP library must be built).

Deliberate differences, all outside the numbers the drivers use:
  * `t.Rij` / `t.Rji` are not produced (the reference accumulates them without ever zeroing or
    reading them, rh_method.py:691-692); accessing them raises AttributeError.
  * collisional rates are evaluated when the Context is built and whenever
    `update_collisions()` is called, not on every formal solution (they depend on the
    atmosphere only, rh_method.py:474-487).
"""
from typing import List, Optional

import numpy as np

from . import _capi, lineprofile
from . import constants as Const
from .problem import Problem, Transition, ColumnBlock, Engine


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


class ComputationalTransition:
    """rh_method.py:25-288 (state only; uv() lives in the sweep kernel)."""

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
        self.compute_phi(atmos)                                                        # :123
        self.active = np.zeros(spect.wavelength.shape[0], dtype=bool)                  # :124-127
        for la, s in enumerate(spect.activeSet):
            if _contains(s, trans):
                self.active[la] = True
        self.gij = None

    def lt(self, la: int) -> int:
        return la - self.Nblue

    def wlambda(self, la: Optional[int] = None):
        w = lineprofile.wlambda(self.wavelength, self.lambda0 if self.isLine else None)
        return w if la is None else w[la]

    def compute_phi(self, atmos):
        """rh_method.py:198-243"""
        if not self.isLine:
            return
        aDamp, _ = self.transModel.damping(atmos, self.atom.vBroad, self.atom.hPops.n[0])
        self.aDamp = np.asarray(aDamp, dtype=np.float64)
        self.phi, self.wphi = lineprofile.compute_phi(self.wavelength, self.lambda0, self.aDamp, self.atom.vBroad,
                                                      atmos.vlos, atmos.muz, atmos.wmu)


class ComputationalAtom:
    """rh_method.py:290-487."""

    def __init__(self, atom, atmos, spect, eqPops):
        self.atomicModel = atom
        self.atomicTable = getattr(eqPops, 'atomicTable', None)
        self.spect = spect
        self.atmos = atmos
        self.vBroad = np.asarray(atom.v_broad(atmos), dtype=np.float64)
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
        self.Gamma = np.zeros((Nlevel, Nlevel, atmos.Nspace))
        self.C = np.zeros((Nlevel, Nlevel, atmos.Nspace))
        self.nStar = self.pops.nStar
        if self.pops.pops is not None:                         # warm start, :412-416
            self.n = self.pops.pops
        else:
            self.n = np.copy(self.nStar)
            self.pops.pops = self.n
        self.compute_collisions()

    def compute_collisions(self):
        """rh_method.py:474-487"""
        self.C = np.zeros_like(self.Gamma)
        for col in self.atomicModel.collisions:
            col.compute_rates(self.atmos, self.nStar, self.C)
        self.C[self.C < 0.0] = 0.0


class Context:
    """rh_method.py:490-745 on the GPU."""

    def __init__(self, atmos, spect, eqPops, background, device: int = 0, stream=None, lib=None):
        self.atmos = atmos
        self.atmos.nondimensionalise()
        self.spect = spect
        self.background = background
        self.eqPops = eqPops
        self.activeAtoms: List[ComputationalAtom] = [ComputationalAtom(a, atmos, spect, eqPops)
                                                     for a in spect.radSet.activeAtoms]
        Nspect, Nspace, Nrays = spect.wavelength.shape[0], atmos.Nspace, atmos.Nrays
        self.J = np.zeros((Nspect, Nspace))
        self.I = np.zeros((Nspect, Nrays))

        # ---- flatten to the lsx problem description --------------------------------------
        trans, active = [], []
        for a, atom in enumerate(self.activeAtoms):
            for t in atom.trans:
                tr = Transition(atom=a, is_line=t.isLine, i=t.i, j=t.j, Nblue=t.Nblue, Nlambda=t.wavelength.shape[0])
                if t.isLine:
                    tr.Aji, tr.Bji, tr.Bij, tr.lambda0 = t.Aji, t.Bji, t.Bij, t.lambda0
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
        self._upload()
        self._n_synced = self._cat_n()
        self._J_synced = self.J.copy()

    # -- packing -------------------------------------------------------------------------
    def _cat_n(self):
        return np.concatenate([np.asarray(a.n, dtype=np.float64) for a in self.activeAtoms], axis=0)

    def _block(self) -> ColumnBlock:
        p, atmos, bg = self.problem, self.atmos, self.background
        lines = [t for a in self.activeAtoms for t in a.trans if t.isLine]
        if p.phi_compact:
            phi = np.concatenate([t.phi[:, 0, 0, :] for t in lines], axis=0) if lines else np.zeros((0, p.Nspace))
        else:
            phi = np.concatenate([t.phi for t in lines], axis=0) if lines else np.zeros((0, p.Nrays, 2, p.Nspace))
        wphi = np.stack([t.wphi for t in lines]) if lines else np.zeros((0, p.Nspace))
        sca = np.asarray(bg.sca, dtype=np.float64)
        sca = sca if p.sca_per_lambda else (sca[0] if sca.ndim == 2 else sca)
        return ColumnBlock(
            height=np.asarray(atmos.height, dtype=np.float64)[None], temperature=np.asarray(atmos.temperature, dtype=np.float64)[None],
            nStar=np.concatenate([np.asarray(a.nStar) for a in self.activeAtoms], axis=0)[None],
            nTotal=np.stack([np.asarray(a.nTotal) for a in self.activeAtoms])[None],
            n=self._cat_n()[None],
            C=np.concatenate([a.C.reshape(-1, p.Nspace) for a in self.activeAtoms], axis=0)[None],
            bg_chi=np.asarray(bg.chi, dtype=np.float64)[None], bg_eta=np.asarray(bg.eta, dtype=np.float64)[None],
            bg_sca=sca[None], phi=np.ascontiguousarray(phi)[None], wphi=wphi[None])

    def _upload(self):
        self._engine.set_columns(0, self._block())

    def update_collisions(self):
        """re-evaluate the collisional rates from the atmosphere (the reference does this on every
        formal solution, rh_method.py:589) and send them to the device"""
        for a in self.activeAtoms:
            a.compute_collisions()
        J = self.J.copy()
        self._upload()
        self._engine.set(_capi.LSX_J, J[None])

    def _push_host_edits(self):
        # the reference's arrays are live numpy objects the caller may edit between calls
        n = self._cat_n()
        if not np.array_equal(n, self._n_synced):
            self._engine.set(_capi.LSX_N, n[None])
            self._n_synced = n
        if not np.array_equal(self.J, self._J_synced):
            self._engine.set(_capi.LSX_J, self.J[None])

    # -- the two verbs ---------------------------------------------------------------------
    def formal_sol_gamma_matrices(self) -> float:
        """rh_method.py:565-708 -> max relative change of J"""
        self._push_host_edits()
        dJ = self._engine.formal_sol_gamma()
        self.J[...] = self._engine.get(_capi.LSX_J)[0]
        self.I[...] = self._engine.get(_capi.LSX_I)[0]
        self._J_synced = self.J.copy()
        G = self._engine.get(_capi.LSX_GAMMA)
        for a, atom in enumerate(self.activeAtoms):
            atom.Gamma[...] = self._engine.gamma_of_atom(G, a)[0]
        return dJ

    def stat_equil(self) -> float:
        """rh_method.py:710-745 -> max relative population change; populations are written back IN
        PLACE into the arrays that alias eqPops[...].pops (rh_method.py:412-416, response_fn.py:62)"""
        self._push_host_edits()
        dPops = self._engine.stat_equil()
        n = self._engine.get(_capi.LSX_N)[0]
        off = 0
        for atom in self.activeAtoms:
            atom.n[...] = n[off:off + atom.Nlevel]
            off += atom.Nlevel
        self._n_synced = n
        return dPops

    def close(self):
        self._engine.close()
