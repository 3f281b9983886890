"""ctypes binding of the lsx C ABI (include/lsx.h).

The product loads exactly one library: lightspinner_amd/csrc/liblsx_hip.so (HIP
kernels for gfx950).  There is no CPU fallback: if that library is missing or does
not load, `load_hip_library()` raises.  `LsxLibrary(path)` is generic over the path
only so that tests can bind the oracle (same ABI) as a checker.
"""
import ctypes as C
import os

import numpy as np

ABI_VERSION = 1

# item selectors (include/lsx.h)
LSX_I, LSX_J, LSX_N, LSX_GAMMA, LSX_DJ_COL, LSX_DPOPS_COL, LSX_NSTAR, LSX_C, _, _, LSX_PHI, LSX_WPHI, LSX_VBROAD, LSX_ADAMP = range(14)
LSX_SOLVER_LINEAR, LSX_SOLVER_PARABOLIC = 0, 1
LSX_SWEEP_AUTO, LSX_SWEEP_RAY_PER_LANE, LSX_SWEEP_RAY_SERIAL = 0, 1, 2      # include/lsx.h: lsx_set_sweep_policy
LSX_COLL_OMEGA, LSX_COLL_CI, LSX_COLL_CE = range(3)

LSX_EINVAL, LSX_EDEVICE, LSX_ESINGULAR, LSX_EUNSUPPORTED = 1, 2, 3, 5
ERRORS = {1: 'LSX_EINVAL', 2: 'LSX_EDEVICE', 3: 'LSX_ESINGULAR', 5: 'LSX_EUNSUPPORTED'}

_dp = C.POINTER(C.c_double)


class LsxTransition(C.Structure):
    _fields_ = [('atom', C.c_int32), ('is_line', C.c_int32), ('i', C.c_int32), ('j', C.c_int32),
                ('Nblue', C.c_int32), ('Nlambda', C.c_int32),
                ('Aji', C.c_double), ('Bji', C.c_double), ('Bij', C.c_double), ('lambda0', C.c_double)]


class LsxProblem(C.Structure):
    _fields_ = [('abi_version', C.c_int32),
                ('Nspace', C.c_int32), ('Nrays', C.c_int32), ('Nspect', C.c_int32),
                ('Natoms', C.c_int32), ('Ntrans', C.c_int32),
                ('Nlevel', C.POINTER(C.c_int32)),
                ('wavelength', _dp), ('muz', _dp), ('wmu', _dp),
                ('trans', C.POINTER(LsxTransition)),
                ('active', C.POINTER(C.c_uint8)),
                ('alpha', _dp),
                ('sca_per_lambda', C.c_int32), ('phi_compact', C.c_int32)]


class LsxColumns(C.Structure):
    _fields_ = [(k, _dp) for k in ('height', 'temperature', 'nStar', 'nTotal', 'n', 'C',
                                   'bg_chi', 'bg_eta', 'bg_sca', 'phi', 'wphi')]


class LsxLevel(C.Structure):
    _fields_ = [('E_SI', C.c_double), ('g', C.c_double), ('stage', C.c_int32), ('reserved', C.c_int32)]


class LsxLineModel(C.Structure):
    _fields_ = [('i', C.c_int32), ('j', C.c_int32), ('gRad', C.c_double), ('stark', C.c_double),
                ('vdw_kind', C.c_int32), ('reserved', C.c_int32), ('vdw', C.c_double * 2)]


class LsxCollision(C.Structure):
    _fields_ = [('kind', C.c_int32), ('i', C.c_int32), ('j', C.c_int32), ('nT', C.c_int32),
                ('temperature', _dp), ('rates', _dp)]


class LsxAtomModel(C.Structure):
    _fields_ = [('weight', C.c_double), ('is_hydrogen', C.c_int32), ('Nlevel', C.c_int32),
                ('levels', C.POINTER(LsxLevel)), ('Nline', C.c_int32), ('Ncollision', C.c_int32),
                ('lines', C.POINTER(LsxLineModel)), ('collisions', C.POINTER(LsxCollision))]


class LsxAtomicData(C.Structure):
    _fields_ = [('Natoms', C.c_int32), ('reserved', C.c_int32), ('atoms', C.POINTER(LsxAtomModel)),
                ('weight_H', C.c_double), ('weight_He', C.c_double), ('abundance_He', C.c_double)]


class LsxAtmosphere(C.Structure):
    _fields_ = [(k, _dp) for k in ('temperature', 'ne', 'vturb', 'vlos', 'nHGround', 'nTotal')] + \
               [('lte_pops', C.c_int32), ('reserved', C.c_int32)]


class LsxTransGrid(C.Structure):
    _fields_ = [('is_line', C.c_int32), ('n', C.c_int32), ('wavelength', _dp), ('lambdaEdge', C.c_double)]


class LsxContinuumModel(C.Structure):
    _fields_ = [('hydrogenic', C.c_int32), ('n', C.c_int32), ('wavelength', _dp), ('alpha', _dp), ('lambdaEdge', C.c_double),
                ('minLambda', C.c_double), ('alpha0', C.c_double), ('E_i', C.c_double), ('E_j', C.c_double),
                ('stage_j', C.c_int32), ('reserved', C.c_int32)]


# every symbol include/lsx.h declares
REQUIRED_SYMBOLS = (
    'lsx_create', 'lsx_destroy', 'lsx_set_columns', 'lsx_formal_sol_gamma', 'lsx_stat_equil',
    'lsx_formal_sol_gamma_async', 'lsx_stat_equil_async', 'lsx_sync', 'lsx_get', 'lsx_set',
    'lsx_piecewise_linear_1d', 'lsx_time_formal_sol', 'lsx_last_error', 'lsx_backend_name',
    'lsx_abi_version', 'lsx_algorithmic_bytes_per_column', 'lsx_set_active_columns', 'lsx_set_line_profiles',
    'lsx_piecewise_1d_impl', 'lsx_w2', 'lsx_monitors', 'lsx_set_atomic_data', 'lsx_set_atmosphere',
    'lsx_wavelength_grid', 'lsx_active_set', 'lsx_line_wavelength', 'lsx_continuum_alpha',
    'lsx_piecewise_parabolic_1d_impl', 'lsx_w3', 'lsx_set_formal_solver',
    'lsx_sync_begin', 'lsx_sync_end', 'lsx_formal_sol_gamma_speculative', 'lsx_discard_formal_sol', 'lsx_prefers_lookahead',
    'lsx_set_sweep_policy', 'lsx_sweep_policy',
    'lsx_create_with_options', 'lsx_effective_options', 'lsx_options_signature',
    'lsx_sync_begin_populations', 'lsx_fetch_populations', 'lsx_build_id',
)


class LsxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('%s: %s' % (ERRORS.get(code, 'error %d' % code), msg))
        self.code = code


class LsxSingularError(LsxError, np.linalg.LinAlgError):
    """Singular statistical-equilibrium system; the reference raises
    numpy.linalg.LinAlgError from scipy.linalg.solve here (rh_method.py:739)."""


def _ptr(a):
    return a.ctypes.data_as(_dp)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError('expected shape %s, got %s' % (tuple(shape), a.shape))
    return a


class LsxLibrary:
    """A loaded library exporting the lsx ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.dll = C.CDLL(path, mode=getattr(os, 'RTLD_LOCAL', 0) | getattr(os, 'RTLD_NOW', 2))
        missing = [s for s in REQUIRED_SYMBOLS if not hasattr(self.dll, s)]
        # (profiles/ab.sh compares library variants built from older sources on one box: LSX_AB_OLD_ABI=1 lets a variant without the
        # entries of the round after it load (round 6: the populations' read-back and the build id); Engine then creates its contexts
        # with plain lsx_create)
        self.old_abi = bool(missing) and os.environ.get('LSX_AB_OLD_ABI') == '1' and set(missing) <= set(REQUIRED_SYMBOLS[-3:])
        if missing and not self.old_abi:
            raise ImportError('%s does not export: %s' % (path, ', '.join(missing)))
        d = self.dll
        d.lsx_abi_version.restype = C.c_int32
        if d.lsx_abi_version() != ABI_VERSION:
            raise ImportError('%s: ABI version %d, expected %d' % (path, d.lsx_abi_version(), ABI_VERSION))
        d.lsx_last_error.restype = C.c_char_p
        d.lsx_backend_name.restype = C.c_char_p
        d.lsx_create.argtypes = [C.POINTER(LsxProblem), C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
        d.lsx_destroy.argtypes = [C.c_void_p]
        d.lsx_destroy.restype = None
        d.lsx_set_columns.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(LsxColumns)]
        d.lsx_formal_sol_gamma.argtypes = [C.c_void_p, _dp]
        d.lsx_stat_equil.argtypes = [C.c_void_p, _dp]
        d.lsx_formal_sol_gamma_async.argtypes = [C.c_void_p]
        d.lsx_stat_equil_async.argtypes = [C.c_void_p]
        d.lsx_sync.argtypes = [C.c_void_p, _dp, _dp]
        d.lsx_get.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _dp, C.c_size_t]
        d.lsx_set.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _dp, C.c_size_t]
        d.lsx_piecewise_linear_1d.argtypes = [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp,
                                              C.POINTER(C.c_int32), _dp, _dp, _dp, _dp, _dp]
        d.lsx_piecewise_1d_impl.argtypes = [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, C.POINTER(C.c_int32), _dp, _dp, _dp,
                                            _dp, _dp]
        d.lsx_w2.argtypes = [C.c_int32, C.c_int32, _dp, _dp]
        d.lsx_w3.argtypes = [C.c_int32, C.c_int32, _dp, _dp]
        d.lsx_piecewise_parabolic_1d_impl.argtypes = [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, C.POINTER(C.c_int32), _dp, _dp, _dp,
                                                      _dp, _dp]
        d.lsx_set_formal_solver.argtypes = [C.c_void_p, C.c_int32]
        d.lsx_set_sweep_policy.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        d.lsx_sweep_policy.argtypes = [C.c_void_p]
        if not self.old_abi:
            d.lsx_create_with_options.argtypes = [C.POINTER(LsxProblem), C.c_int32, C.c_int32, C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
            d.lsx_effective_options.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
            d.lsx_options_signature.argtypes = [C.c_void_p]
            d.lsx_options_signature.restype = C.c_uint64
        d.lsx_sweep_policy.restype = C.c_int32
        ip = C.POINTER(C.c_int32)
        d.lsx_wavelength_grid.argtypes = [C.c_int32, C.POINTER(LsxTransGrid), C.c_int32, _dp, C.c_double, C.c_int32, _dp, ip, ip, ip]
        d.lsx_active_set.argtypes = [C.c_int32, C.c_int32, ip, ip, C.POINTER(C.c_uint8)]
        d.lsx_line_wavelength.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, _dp, ip]
        d.lsx_continuum_alpha.argtypes = [C.POINTER(LsxContinuumModel), C.c_int32, _dp, _dp]
        d.lsx_monitors.argtypes = [C.c_void_p, C.c_void_p]
        d.lsx_sync_begin.argtypes = [C.c_void_p]
        d.lsx_sync_end.argtypes = [C.c_void_p, _dp, _dp]
        if not self.old_abi:          # (round 6's entries; a round-5 library loaded for an A/B lacks exactly these three)
            d.lsx_sync_begin_populations.argtypes = [C.c_void_p]
            d.lsx_fetch_populations.argtypes = [C.c_void_p, _dp, C.c_size_t]
            d.lsx_build_id.restype = C.c_char_p
        d.lsx_formal_sol_gamma_speculative.argtypes = [C.c_void_p]
        d.lsx_discard_formal_sol.argtypes = [C.c_void_p]
        d.lsx_prefers_lookahead.argtypes = [C.c_void_p]
        d.lsx_set_atomic_data.argtypes = [C.c_void_p, C.POINTER(LsxAtomicData)]
        d.lsx_set_atmosphere.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(LsxAtmosphere)]
        d.lsx_set_active_columns.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        d.lsx_set_line_profiles.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _dp]
        d.lsx_time_formal_sol.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp]
        d.lsx_algorithmic_bytes_per_column.argtypes = [C.c_void_p]
        d.lsx_algorithmic_bytes_per_column.restype = C.c_double

    @property
    def backend(self):
        return self.dll.lsx_backend_name().decode()

    def check(self, rc):
        if rc != 0:
            msg = self.dll.lsx_last_error().decode(errors='replace')
            if rc == 3:
                raise LsxSingularError(rc, msg)
            raise LsxError(rc, msg)

    def piecewise_linear_1d(self, height, temperature, mu, to_obs, wav, chi, S, device=0):
        """Batched formal_solver.piecewise_linear_1d (formal_solver.py:144-212)."""
        chi = f64(chi)
        S = f64(S, chi.shape)
        if chi.ndim != 2:
            raise ValueError('chi, S must be [nray, Nspace]')
        nray, ns = chi.shape
        height = f64(height, (ns,))
        temperature = f64(temperature, (ns,))
        mu = f64(mu, (nray,))
        wav = f64(wav, (nray,))
        to_obs = np.ascontiguousarray(to_obs, dtype=np.int32)
        if to_obs.shape != (nray,):
            raise ValueError('to_obs must be [nray]')
        I = np.empty_like(chi)
        Psi = np.empty_like(chi)
        self.check(self.dll.lsx_piecewise_linear_1d(device, nray, ns, _ptr(height), _ptr(temperature), _ptr(mu),
                                                    to_obs.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(wav),
                                                    _ptr(chi), _ptr(S), _ptr(I), _ptr(Psi)))
        return I, Psi


    def piecewise_parabolic_1d_impl(self, height, mu, to_obs, Istart, chi, S, device=0):
        """Batched monotonic piecewise-parabolic short characteristics (include/lsx.h, N4); arguments as piecewise_1d_impl."""
        return self.piecewise_1d_impl(height, mu, to_obs, Istart, chi, S, device=device, _entry='lsx_piecewise_parabolic_1d_impl')

    def piecewise_1d_impl(self, height, mu, to_obs, Istart, chi, S, device=0, _entry='lsx_piecewise_1d_impl'):
        """Batched formal_solver.piecewise_1d_impl (formal_solver.py:46-142): incident intensity handed over."""
        chi = f64(chi)
        S = f64(S, chi.shape)
        if chi.ndim != 2:
            raise ValueError('chi, S must be [nray, Nspace]')
        nray, ns = chi.shape
        height = f64(height, (ns,))
        mu = f64(mu, (nray,))
        Istart = f64(Istart, (nray,))
        to_obs = np.ascontiguousarray(to_obs, dtype=np.int32)
        if to_obs.shape != (nray,):
            raise ValueError('to_obs must be [nray]')
        I = np.empty_like(chi)
        Psi = np.empty_like(chi)
        self.check(getattr(self.dll, _entry)(device, nray, ns, _ptr(height), _ptr(mu), to_obs.ctypes.data_as(C.POINTER(C.c_int32)),
                                             _ptr(Istart), _ptr(chi), _ptr(S), _ptr(I), _ptr(Psi)))
        return I, Psi

    # -- wavelength grid and active set (host side of the library) --------------------------------------------------
    def wavelength_grid(self, grids, is_line, edges, extra=None, lambda_reference=500.0):
        """RadiativeSet.compute_wavelength_grid's merge (atomic_set.py:377-416): grids = the transitions' own wavelength
        arrays -> (wavelength [Nspect], blueIdx [Ntrans], redIdx [Ntrans])"""
        n = len(grids)
        keep = [f64(g).reshape(-1) for g in grids]
        tg = (LsxTransGrid * max(1, n))()
        for q in range(n):
            tg[q] = LsxTransGrid(1 if is_line[q] else 0, keep[q].shape[0], _ptr(keep[q]), float(edges[q]) if not is_line[q] else 0.0)
        ex = f64(extra).reshape(-1) if extra is not None else np.zeros(0)
        cap = int(sum(k.shape[0] + 1 for k in keep) + ex.shape[0] + 1)
        wav = np.empty(cap)
        ns = C.c_int32()
        blue, red = np.zeros(max(1, n), dtype=np.int32), np.zeros(max(1, n), dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        self.check(self.dll.lsx_wavelength_grid(n, tg, ex.shape[0], _ptr(ex), float(lambda_reference), cap, _ptr(wav), C.byref(ns),
                                                blue.ctypes.data_as(ip), red.ctypes.data_as(ip)))
        return wav[:ns.value].copy(), blue[:n], red[:n]

    def active_set(self, blue, red, Nspect):
        """-> bool [Ntrans][Nspect]: lsx_problem.active / the membership of spect.activeSet (atomic_set.py:418-453)"""
        blue, red = np.ascontiguousarray(blue, dtype=np.int32), np.ascontiguousarray(red, dtype=np.int32)
        act = np.zeros((blue.shape[0], int(Nspect)), dtype=np.uint8)
        ip = C.POINTER(C.c_int32)
        self.check(self.dll.lsx_active_set(blue.shape[0], int(Nspect), blue.ctypes.data_as(ip), red.ctypes.data_as(ip),
                                           act.ctypes.data_as(C.POINTER(C.c_uint8))))
        return act.astype(bool)

    def line_wavelength(self, lambda0, qCore, qWing, NlambdaGen):
        """VoigtLine.setup_wavelength (atomic_model.py:347-380)"""
        out = np.empty(int(NlambdaGen) + 2)
        n = C.c_int32()
        self.check(self.dll.lsx_line_wavelength(float(lambda0), float(qCore), float(qWing), int(NlambdaGen), out.shape[0], _ptr(out),
                                                C.byref(n)))
        return out[:n.value].copy()

    def continuum_alpha(self, wavelength, *, edge, min_lambda, table=None, alpha0=0.0, E_i=0.0, E_j=0.0, stage_j=0):
        """compute_alpha (atomic_model.py:606-612 for table = (wavelength, alpha), :662-671 hydrogenic otherwise)"""
        w = f64(wavelength).reshape(-1)
        out = np.empty_like(w)
        m = LsxContinuumModel()
        m.lambdaEdge, m.minLambda = float(edge), float(min_lambda)
        if table is not None:
            x, y = f64(table[0]).reshape(-1), f64(table[1]).reshape(-1)
            if x.shape != y.shape:
                raise ValueError('table: wavelength and alpha differ in length')
            m.hydrogenic, m.n, m.wavelength, m.alpha = 0, x.shape[0], _ptr(x), _ptr(y)
        else:
            m.hydrogenic, m.alpha0, m.E_i, m.E_j, m.stage_j = 1, float(alpha0), float(E_i), float(E_j), int(stage_j)
        self.check(self.dll.lsx_continuum_alpha(C.byref(m), w.shape[0], _ptr(w), _ptr(out)))
        return out

    def w3(self, dtau, device=0):
        """weights of the parabolic rule (include/lsx.h, N4) -> [n][3] (w0, w1, w2)"""
        dtau = f64(dtau).reshape(-1)
        out = np.empty((dtau.shape[0], 3))
        self.check(self.dll.lsx_w3(device, dtau.shape[0], _ptr(dtau), _ptr(out)))
        return out

    def w2(self, dtau, device=0):
        """formal_solver.w2 (formal_solver.py:14-44) on an array -> [n][2] (w0, w1)."""
        dtau = f64(dtau).reshape(-1)
        out = np.empty((dtau.shape[0], 2))
        self.check(self.dll.lsx_w2(device, dtau.shape[0], _ptr(dtau), _ptr(out)))
        return out


_HIP_LIB = None


def hip_library_path():
    """the in-tree product library; LSX_HIP_LIBRARY names another build of it (A/B runs of profiles/ab.sh variants)"""
    return os.environ.get('LSX_HIP_LIBRARY') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'liblsx_hip.so')


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so and ask
    for them by the unversioned name, which the loader does not match against an already loaded /opt/rocm
    libamdhip64.so.7: if this library came first, a later `import torch` in the same process (torch.distributed,
    streams) would bring up a second runtime that finds no GPU.  Loading torch's copy first makes both sides resolve
    to it (same SONAME), whichever is imported first.  Without PyTorch the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    rt = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if not os.path.exists(rt):
        return None
    return C.CDLL(rt, mode=getattr(os, 'RTLD_GLOBAL', 0x100) | getattr(os, 'RTLD_NOW', 2))


def load_hip_library():
    """Load the HIP backend or fail loudly -- the product has no other backend."""
    global _HIP_LIB
    if _HIP_LIB is None:
        path = hip_library_path()
        if not os.path.exists(path):
            raise ImportError(
                'lightspinner_amd: HIP extension %s not built (run `python -c "import __graft_entry__ as g; '
                'g.build()"` or `make -C lightspinner_amd/csrc`). There is no CPU fallback.' % path)
        _share_hip_runtime_with_torch()
        _HIP_LIB = LsxLibrary(path)
        if not _HIP_LIB.backend.startswith('hip'):
            raise ImportError('%s is not the HIP backend (%s)' % (path, _HIP_LIB.backend))
    return _HIP_LIB
