"""Flat, column-batched description of the hot-path inputs and the Engine that
feeds them through the lsx C ABI.

`Problem`      -- what is common to every column: grids, quadrature, transition table
                  (what rh_method.Context.__init__ derives from `spect`, rh_method.py:531-563)
`ColumnBlock`  -- per-column arrays with a leading [ncol] index, in the reference's
                  own layouts (rh_method.py:387-423 and 93-131)
`Engine`       -- owns one lsx_ctx (one device, one stream)
"""
import ctypes as C
import itertools
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _capi
from ._capi import f64, _ptr


@dataclass
class Transition:
    atom: int
    is_line: bool
    i: int
    j: int
    Nblue: int
    Nlambda: int
    Aji: float = 0.0
    Bji: float = 0.0
    Bij: float = 0.0
    lambda0: float = 0.0
    alpha: Optional[np.ndarray] = None  # continua: [Nlambda]


@dataclass
class Problem:
    Nspace: int
    wavelength: np.ndarray          # [Nspect] nm
    muz: np.ndarray                 # [Nrays]
    wmu: np.ndarray                 # [Nrays]
    Nlevel: List[int]               # per active atom
    trans: List[Transition]         # ordered as [t for atom in activeAtoms for t in atom.trans]
    active: np.ndarray              # bool [Ntrans][Nspect]
    sca_per_lambda: bool = False
    phi_compact: bool = False
    atom_names: List[str] = field(default_factory=list)

    def __post_init__(self):
        self.wavelength = f64(self.wavelength)
        self.muz = f64(self.muz)
        self.wmu = f64(self.wmu)
        self.Nlevel = [int(x) for x in self.Nlevel]
        self.active = np.ascontiguousarray(self.active, dtype=np.uint8).reshape(len(self.trans), self.Nspect)
        if self.muz.shape != self.wmu.shape:
            raise ValueError('muz and wmu must have the same shape')

    @property
    def Nspect(self): return int(self.wavelength.shape[0])
    @property
    def Nrays(self): return int(self.muz.shape[0])
    @property
    def Natoms(self): return len(self.Nlevel)
    @property
    def Ntrans(self): return len(self.trans)
    @property
    def NLtot(self): return int(sum(self.Nlevel))
    @property
    def NL2tot(self): return int(sum(n * n for n in self.Nlevel))
    @property
    def lines(self): return [t for t in self.trans if t.is_line]
    @property
    def Nlines(self): return len(self.lines)
    @property
    def SNl(self): return int(sum(t.Nlambda for t in self.lines))
    @property
    def lev_off(self): return np.concatenate([[0], np.cumsum(self.Nlevel)[:-1]]).astype(int)
    @property
    def lev2_off(self): return np.concatenate([[0], np.cumsum([n * n for n in self.Nlevel])[:-1]]).astype(int)

    def phi_shape(self):
        if self.phi_compact:
            return (self.SNl, self.Nspace)
        return (self.SNl, self.Nrays, 2, self.Nspace)

    def sca_shape(self):
        return (self.Nspect, self.Nspace) if self.sca_per_lambda else (self.Nspace,)

    def work_units_per_column(self):
        """depth-points x wavelengths x rays (both directions) per FS call."""
        return self.Nspect * self.Nrays * 2 * self.Nspace

    def to_c(self):
        """-> (LsxProblem, keepalive list)"""
        keep = []
        tarr = (_capi.LsxTransition * max(1, self.Ntrans))()
        alphas = []
        for k, t in enumerate(self.trans):
            tarr[k] = _capi.LsxTransition(t.atom, 1 if t.is_line else 0, t.i, t.j, t.Nblue, t.Nlambda,
                                          t.Aji, t.Bji, t.Bij, t.lambda0)
            if not t.is_line:
                a = f64(t.alpha, (t.Nlambda,))
                alphas.append(a)
        alpha = np.concatenate(alphas) if alphas else np.zeros(1)
        alpha = f64(alpha)
        nlevel = np.ascontiguousarray(self.Nlevel, dtype=np.int32)
        p = _capi.LsxProblem()
        p.abi_version = _capi.ABI_VERSION
        p.Nspace, p.Nrays, p.Nspect = self.Nspace, self.Nrays, self.Nspect
        p.Natoms, p.Ntrans = self.Natoms, self.Ntrans
        p.Nlevel = nlevel.ctypes.data_as(C.POINTER(C.c_int32))
        p.wavelength = _ptr(self.wavelength)
        p.muz = _ptr(self.muz)
        p.wmu = _ptr(self.wmu)
        p.trans = tarr
        p.active = self.active.ctypes.data_as(C.POINTER(C.c_uint8))
        p.alpha = _ptr(alpha)
        p.sca_per_lambda = 1 if self.sca_per_lambda else 0
        p.phi_compact = 1 if self.phi_compact else 0
        keep += [tarr, alpha, nlevel]
        return p, keep


_COLUMN_FIELDS = ('height', 'temperature', 'nStar', 'nTotal', 'n', 'C', 'bg_chi', 'bg_eta', 'bg_sca', 'phi', 'wphi')


@dataclass
class ColumnBlock:
    """Per-column inputs, leading index = column (include/lsx.h: lsx_columns)."""
    height: np.ndarray
    temperature: np.ndarray
    nStar: np.ndarray
    nTotal: np.ndarray
    n: np.ndarray
    C: np.ndarray
    bg_chi: np.ndarray
    bg_eta: np.ndarray
    bg_sca: np.ndarray
    phi: Optional[np.ndarray] = None      # None (with wphi None): Engine.set_line_profiles computes them on the device
    wphi: Optional[np.ndarray] = None

    @property
    def ncol(self):
        return int(self.height.shape[0])

    def validate(self, p: Problem):
        nc, Ns = self.ncol, p.Nspace
        shapes = dict(height=(nc, Ns), temperature=(nc, Ns), nStar=(nc, p.NLtot, Ns), nTotal=(nc, p.Natoms, Ns),
                      n=(nc, p.NLtot, Ns), C=(nc, p.NL2tot, Ns), bg_chi=(nc, p.Nspect, Ns),
                      bg_eta=(nc, p.Nspect, Ns), bg_sca=(nc,) + p.sca_shape(), phi=(nc,) + p.phi_shape(),
                      wphi=(nc, p.Nlines, Ns))
        if (self.phi is None) != (self.wphi is None):
            raise ValueError('phi and wphi must both be given or both be None')
        for k in _COLUMN_FIELDS:
            if getattr(self, k) is not None:
                setattr(self, k, f64(getattr(self, k), shapes[k]))
        return self

    def slice(self, c0, c1):
        return ColumnBlock(**{k: (None if getattr(self, k) is None else getattr(self, k)[c0:c1]) for k in _COLUMN_FIELDS})

    def to_c(self):
        s = _capi.LsxColumns()
        for k in _COLUMN_FIELDS:
            if getattr(self, k) is not None:
                setattr(s, k, _ptr(getattr(self, k)))
        return s

    @staticmethod
    def concatenate(blocks):
        return ColumnBlock(**{k: (None if getattr(blocks[0], k) is None else np.concatenate([getattr(b, k) for b in blocks]))
                              for k in _COLUMN_FIELDS})


class Engine:
    """One lsx_ctx.  `lib=None` binds the HIP backend (and raises if it is not built)."""
    _serials = itertools.count(1)

    def __init__(self, problem: Problem, ncol: int, device: int = 0, stream: Optional[int] = None, lib=None,
                 policy_columns: Optional[int] = None, sweep_policy: str = 'auto', options=None):
        """policy_columns: the column count of the WHOLE problem this engine holds a shard of (None: its own `ncol`).  The HIP
        library picks its sweep kernel by a column count; a driver that splits N columns over several engines passes N to all
        of them, so that every column gets the bits it gets when all N sit in one engine (include/lsx.h, lsx_set_sweep_policy).
        options: explicit plan / runtime switches, "key=value,..." or a dict (include/lsx.h, lsx_create_with_options); what the
        engine ended up with: effective_options() / options_signature()."""
        self.lib = lib if lib is not None else _capi.load_hip_library()
        self.serial = next(Engine._serials)      # this process's n-th engine: what per-engine bookkeeping keys on (never re-used, unlike id())
        self.problem = problem
        self.ncol = int(ncol)
        self._h = C.c_void_p()
        cprob, self._keep = problem.to_c()
        if isinstance(options, dict):
            options = ','.join('%s=%s' % (k, int(v) if isinstance(v, bool) else v) for k, v in options.items())
        if getattr(self.lib, 'old_abi', False) and not options:
            self.lib.check(self.lib.dll.lsx_create(C.byref(cprob), self.ncol, int(device), C.c_void_p(stream) if stream else None, C.byref(self._h)))
        else:
            self.lib.check(self.lib.dll.lsx_create_with_options(C.byref(cprob), self.ncol, int(device), C.c_void_p(stream) if stream else None,
                                                                options.encode() if options else None, C.byref(self._h)))
        if policy_columns is not None or sweep_policy != 'auto':
            self.set_sweep_policy(sweep_policy, policy_columns)

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.dll.lsx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data in --------------------------------------------------------------
    def set_columns(self, col0: int, block: ColumnBlock):
        block.validate(self.problem)
        cs = block.to_c()
        self.lib.check(self.lib.dll.lsx_set_columns(self._h, int(col0), block.ncol, C.byref(cs)))

    def set_line_profiles(self, col0, aDamp, vBroad, vlos=None):
        """ComputationalTransition.compute_phi (rh_method.py:198-243) on the device, for columns
        [col0, col0 + ncol): aDamp [ncol][Nlines][Nspace], vBroad [ncol][Natoms][Nspace], vlos [ncol][Nspace]."""
        p = self.problem
        aDamp = f64(aDamp)
        ncol = aDamp.shape[0]
        aDamp = f64(aDamp, (ncol, p.Nlines, p.Nspace))
        vBroad = f64(vBroad, (ncol, p.Natoms, p.Nspace))
        vl = None if vlos is None else f64(vlos, (ncol, p.Nspace))
        self.lib.check(self.lib.dll.lsx_set_line_profiles(self._h, int(col0), ncol, _ptr(aDamp), _ptr(vBroad),
                                                          _ptr(vl) if vl is not None else None))

    def set_atomic_data(self, data):
        """atomdata.AtomicData -> lsx_set_atomic_data (once per engine, before set_atmosphere)"""
        cd, keep = data.to_c()
        self.lib.check(self.lib.dll.lsx_set_atomic_data(self._h, C.byref(cd)))
        del keep

    def set_atmosphere(self, col0, temperature, ne, vturb, nHGround, nTotal, vlos=None, lte_pops=False):
        """lsx_set_atmosphere for columns [col0, col0 + ncol): the library derives vBroad, aDamp, the line profiles, the
        collisional rates and (lte_pops) the LTE populations from the atmosphere (rh_method.py:198-243, 474-487;
        atomic_model.py:66-69, 491-502; atomic_set.py:105-145).  Arrays [ncol][Nspace]; nTotal [ncol][Natoms][Nspace]."""
        p = self.problem
        T = f64(temperature)
        ncol = T.shape[0]
        T = f64(T, (ncol, p.Nspace))
        arrs = dict(temperature=T, ne=f64(ne, (ncol, p.Nspace)), vturb=f64(vturb, (ncol, p.Nspace)),
                    nHGround=f64(nHGround, (ncol, p.Nspace)), nTotal=f64(nTotal, (ncol, p.Natoms, p.Nspace)))
        if vlos is not None:
            arrs['vlos'] = f64(vlos, (ncol, p.Nspace))
        a = _capi.LsxAtmosphere()
        for k, v in arrs.items():
            setattr(a, k, _ptr(v))
        a.lte_pops = 1 if lte_pops else 0
        self.lib.check(self.lib.dll.lsx_set_atmosphere(self._h, int(col0), ncol, C.byref(a)))

    def set(self, what, arr, col0=0):
        arr = f64(arr)
        ncol = arr.shape[0]
        self.lib.check(self.lib.dll.lsx_set(self._h, what, int(col0), ncol, _ptr(arr), arr.nbytes))

    def set_formal_solver(self, solver):
        """'linear' (the reference's piecewise_linear_1d, default) or 'parabolic' (monotonic piecewise parabolic, include/lsx.h N4)"""
        kind = {'linear': _capi.LSX_SOLVER_LINEAR, 'parabolic': _capi.LSX_SOLVER_PARABOLIC}[solver]
        self.lib.check(self.lib.dll.lsx_set_formal_solver(self._h, kind))

    def set_sweep_policy(self, policy='auto', decide_for_columns=None):
        """'auto' (by column count: `decide_for_columns`, default this engine's own), 'ray-per-lane' or 'ray-serial'"""
        kind = {'auto': _capi.LSX_SWEEP_AUTO, 'ray-per-lane': _capi.LSX_SWEEP_RAY_PER_LANE,
                'ray-serial': _capi.LSX_SWEEP_RAY_SERIAL}[policy]
        self.lib.check(self.lib.dll.lsx_set_sweep_policy(self._h, kind, int(decide_for_columns or 0)))

    def effective_options(self) -> str:
        """everything that decides how this engine associates its sums and launches its kernels: the options it was created with
        (environment defaults + explicit list), the rule, the sweep mapping the policy selects, the plan's class list"""
        for size in (4096, 1 << 16, 1 << 20):        # (the class list grows with the plan: include/lsx.h)
            buf = C.create_string_buffer(size)
            if self.lib.dll.lsx_effective_options(self._h, buf, size) == 0:
                return buf.value.decode()
        self.lib.check(self.lib.dll.lsx_effective_options(self._h, buf, size))
        return buf.value.decode()

    def options_signature(self) -> int:
        """64-bit hash of effective_options(): engines with equal signatures on equal problems give every column the same bits"""
        return int(self.lib.dll.lsx_options_signature(self._h))

    def sweep_policy(self) -> str:
        """the mapping the next formal solution runs -- under the parabolic rule: for the classes that have a ray-serial instance of it
        ('oracle' for the CPU restatement, which has one code path)"""
        return {0: 'oracle', 1: 'ray-per-lane', 2: 'ray-serial'}[int(self.lib.dll.lsx_sweep_policy(self._h))]

    def set_active_columns(self, mask=None):
        """freeze columns whose mask entry is False (None: all active)"""
        if mask is None:
            self.lib.check(self.lib.dll.lsx_set_active_columns(self._h, None))
            return
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        if m.shape != (self.ncol,):
            raise ValueError('mask must have one entry per column')
        self.lib.check(self.lib.dll.lsx_set_active_columns(self._h, m.ctypes.data_as(C.POINTER(C.c_uint8))))

    # -- hot path -------------------------------------------------------------
    def formal_sol_gamma(self) -> float:
        v = C.c_double()
        self.lib.check(self.lib.dll.lsx_formal_sol_gamma(self._h, C.byref(v)))
        return v.value

    def stat_equil(self) -> float:
        v = C.c_double()
        self.lib.check(self.lib.dll.lsx_stat_equil(self._h, C.byref(v)))
        return v.value

    def formal_sol_gamma_async(self):
        self.lib.check(self.lib.dll.lsx_formal_sol_gamma_async(self._h))

    def stat_equil_async(self):
        self.lib.check(self.lib.dll.lsx_stat_equil_async(self._h))

    def sync(self):
        a, b = C.c_double(), C.c_double()
        self.lib.check(self.lib.dll.lsx_sync(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- the loop without a host round trip per iteration (include/lsx.h; drivers.iterate_mali_engine) ----
    def sync_begin(self, populations=False):
        """enqueue the read-back of the monitors of the calls enqueued so far (populations: and of n, for fetch_populations)"""
        if populations:
            self.lib.check(self.lib.dll.lsx_sync_begin_populations(self._h))
        else:
            self.lib.check(self.lib.dll.lsx_sync_begin(self._h))

    def fetch_populations(self):
        """the populations [ncol][NLtot][Nspace] as the last collected sync_begin(populations=True) read them back -- host to host:
        it does not wait for what has been enqueued behind that read-back"""
        out = np.empty((self.ncol,) + self._shape(_capi.LSX_N), dtype=np.float64)
        self.lib.check(self.lib.dll.lsx_fetch_populations(self._h, _ptr(out), out.nbytes))
        return out

    def sync_end(self):
        """-> (dJ, dPops) of that read-back; what was enqueued behind it keeps running"""
        a, b = C.c_double(), C.c_double()
        self.lib.check(self.lib.dll.lsx_sync_end(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def formal_sol_gamma_speculative(self):
        """the next iteration's formal solution, enqueued before the monitors of this one are known"""
        self.lib.check(self.lib.dll.lsx_formal_sol_gamma_speculative(self._h))

    def prefers_lookahead(self) -> bool:
        """whether enqueueing ahead pays for this context (include/lsx.h, lsx_prefers_lookahead)"""
        return bool(self.lib.dll.lsx_prefers_lookahead(self._h))

    def discard_formal_sol(self):
        """take the speculative formal solution back: I, J, Gamma and the monitors are the previous call's again"""
        self.lib.check(self.lib.dll.lsx_discard_formal_sol(self._h))

    def monitors_to(self, ptr):
        """lsx_monitors: (max dJ, max dPops, NaN flag, singular flag) of the enqueued calls -> 4 doubles at `ptr`
        (an int address: device memory for the HIP library -- e.g. tensor.data_ptr() -- host memory for the oracle)"""
        self.lib.check(self.lib.dll.lsx_monitors(self._h, C.c_void_p(int(ptr))))

    def time_formal_sol(self, warmup, reps):
        a, b = C.c_double(), C.c_double()
        self.lib.check(self.lib.dll.lsx_time_formal_sol(self._h, int(warmup), int(reps), C.byref(a), C.byref(b)))
        return a.value, b.value

    def algorithmic_bytes_per_column(self) -> float:
        return float(self.lib.dll.lsx_algorithmic_bytes_per_column(self._h))

    # -- data out -------------------------------------------------------------
    def _shape(self, what):
        p = self.problem
        return {_capi.LSX_I: (p.Nspect, p.Nrays), _capi.LSX_J: (p.Nspect, p.Nspace),
                _capi.LSX_N: (p.NLtot, p.Nspace), _capi.LSX_GAMMA: (p.NL2tot, p.Nspace),
                _capi.LSX_DJ_COL: (), _capi.LSX_DPOPS_COL: (), _capi.LSX_NSTAR: (p.NLtot, p.Nspace),
                _capi.LSX_C: (p.NL2tot, p.Nspace), _capi.LSX_PHI: p.phi_shape(),
                _capi.LSX_WPHI: (p.Nlines, p.Nspace), _capi.LSX_VBROAD: (p.Natoms, p.Nspace),
                _capi.LSX_ADAMP: (max(1, p.Nlines), p.Nspace)}[what]

    def get(self, what, col0=0, ncol=None):
        ncol = self.ncol - col0 if ncol is None else ncol
        out = np.empty((ncol,) + self._shape(what), dtype=np.float64)
        self.lib.check(self.lib.dll.lsx_get(self._h, what, int(col0), int(ncol), _ptr(out), out.nbytes))
        return out

    def gamma_of_atom(self, G, a):
        """view [ncol][Nl][Nl][Nspace] of atom a inside an LSX_GAMMA array"""
        p = self.problem
        nl = p.Nlevel[a]
        o = p.lev2_off[a]
        return G[:, o:o + nl * nl, :].reshape(G.shape[0], nl, nl, p.Nspace)
