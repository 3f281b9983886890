"""Run under AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_sanitized_host.py starts this file in a child
process with libasan preloaded).  Two legs:
  plan   the product's host-side code -- lsx_plan.cpp (tile schedule, slot table, classes, strides, launch shapes) and
         lsx_grid.cpp (wavelength grid, active set, line grids, continuum cross-sections) -- built from the product's own
         sources into liblsx_host_asan.so; every plan is also checked against its invariants (lsx_plan_capi.cpp)
  oracle the C restatement (make -C oracle asan) through formal solutions, statistical equilibrium, the parabolic rule,
         the set-up chain and the grid entries
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np

from lightspinner_amd import _capi, fixtures
from lightspinner_amd.problem import Problem, Transition
from conftest import golden


class HostOnly(_capi.LsxLibrary):
    """the sanitizer build of the product's host-only translation units: grid entries + lsx_plan_probe"""

    def __init__(self, path):
        self.path = path
        self.dll = d = C.CDLL(path)
        _dp, ip = _capi._dp, C.POINTER(C.c_int32)
        d.lsx_last_error.restype = C.c_char_p
        d.lsx_wavelength_grid.argtypes = [C.c_int32, C.POINTER(_capi.LsxTransGrid), C.c_int32, _dp, C.c_double, C.c_int32, _dp, ip, ip, ip]
        d.lsx_active_set.argtypes = [C.c_int32, C.c_int32, ip, ip, C.POINTER(C.c_uint8)]
        d.lsx_line_wavelength.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, _dp, ip]
        d.lsx_continuum_alpha.argtypes = [C.POINTER(_capi.LsxContinuumModel), C.c_int32, _dp, _dp]
        d.lsx_plan_probe.argtypes = [C.POINTER(_capi.LsxProblem), C.c_uint32, C.POINTER(C.c_int64), ip, C.c_int32, C.c_char_p, C.c_int32]

    def signature(self, prob, options=None):
        """-> (hash, string): what the product's lsx_options_signature / lsx_effective_options report as far as the host side decides
        it (options from the environment + the explicit list, the plan's class list)"""
        p, keep = prob.to_c()
        f = self.dll.lsx_plan_signature
        f.restype = C.c_uint64
        f.argtypes = [C.POINTER(_capi.LsxProblem), C.c_char_p, C.c_char_p, C.c_int32]
        buf = C.create_string_buffer(4096)
        h = f(C.byref(p), options.encode() if options else None, buf, 4096)
        return int(h), buf.value.decode()

    def probe(self, prob, bits=0):
        """-> (rc, message, summary[16], tiles[ntile][8])"""
        p, keep = prob.to_c()
        summary = (C.c_int64 * 16)()
        tiles = np.zeros((4096, 8), dtype=np.int32)
        err = C.create_string_buffer(512)
        rc = self.dll.lsx_plan_probe(C.byref(p), bits, summary, tiles.ctypes.data_as(C.POINTER(C.c_int32)), 4096, err, 512)
        s = np.array(summary[:], dtype=np.int64)
        return rc, err.value.decode(), s, tiles[:int(s[0])] if rc == 0 else tiles[:0]


SUMMARY = ('tiles', 'slots', 'classes', 'phi_col', 'corr_col', 'pp_col', 'til_col', 'lds_bytes', 'static_max', 'nF_max', 'Ncont',
           'generic_tiles', 'fast_tiles', 'linked_tiles', 'per_ray_slots', 'lane_fill_permille')


def random_problem(rng):
    """a random transition table: 1-3 atoms, lines and continua with random level pairs and wavelength ranges, some of them
    inactive on part of their range"""
    Nrays = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 64]))
    Nspect = int(rng.integers(3, 260))
    Nspace = int(rng.choice([3, 4, 5, 17, 82, 83, 700, 1500]))
    Natoms = int(rng.integers(1, 4))
    Nlevel = [int(rng.integers(2, 9)) for _ in range(Natoms)]
    wavelength = np.cumsum(rng.uniform(0.01, 3.0, Nspect)) + 20.0
    x, w = np.polynomial.legendre.leggauss(Nrays)
    trans = []
    for a in range(Natoms):
        nl = Nlevel[a]
        for _ in range(int(rng.integers(0, 14))):
            i = int(rng.integers(0, nl - 1))
            j = int(rng.integers(i + 1, nl))
            n = int(rng.integers(2, max(3, Nspect)))
            n = min(n, Nspect)
            blue = int(rng.integers(0, Nspect - n + 1))
            if rng.uniform() < 0.6:
                trans.append(Transition(a, True, i, j, blue, n, Aji=1e7, Bji=2e9, Bij=1e9, lambda0=float(wavelength[blue + n // 2])))
            else:
                trans.append(Transition(a, False, i, j, blue, n, lambda0=float(wavelength[blue + n - 1]), alpha=rng.uniform(1e-23, 1e-21, n)))
    trans.sort(key=lambda t: (t.atom, not t.is_line))          # lines first inside an atom, like ComputationalAtom.trans
    active = np.zeros((len(trans), Nspect), dtype=np.uint8)
    for k, t in enumerate(trans):
        active[k, t.Nblue:t.Nblue + t.Nlambda] = 1
        if rng.uniform() < 0.2 and t.Nlambda > 4:               # holes: a transition inactive inside its own range
            h = int(rng.integers(1, t.Nlambda - 1))
            active[k, t.Nblue + h] = 0
    return Problem(Nspace=Nspace, wavelength=wavelength, muz=0.5 * x + 0.5, wmu=0.5 * w, Nlevel=Nlevel, trans=trans, active=active,
                   sca_per_lambda=bool(rng.integers(0, 2)), phi_compact=bool(rng.integers(0, 2)))


def leg_plan(path):
    import grid_cases
    from toy import toy_problem
    from test_toy_topologies import CASES
    lib = HostOnly(path)
    grid_cases.reference_grid_bit_exact(lib)
    grid_cases.line_grids_and_continuum_alpha(lib)
    grid_cases.random_and_edge_cases(lib)
    n = 0
    # the reference's own problems: FALC CaII (25 tiles) and FALC Ca + H (66 tiles), ray-dependent and compact profiles,
    # every diagnostic option of the plan
    for name in ('falc_ca.npz', 'falc_cah.npz', 'falc_ca_vlos.npz'):
        for compact in ((False,) if name == 'falc_ca_vlos.npz' else (False, True)):
            prob, _, _ = fixtures.load_problem_npz(golden(name), phi_compact=compact)
            for bits in (0, 1, 2, 4, 8, 16, 1 | 2 | 4 | 8 | 16, 6 << 8):
                rc, msg, s, tiles = lib.probe(prob, bits)
                assert rc == 0, (name, compact, bits, msg)
                n += 1
            rc, msg, s, tiles = lib.probe(prob, 0)
            info = dict(zip(SUMMARY, s.tolist()))
            if name == 'falc_ca.npz':
                assert info['tiles'] == 25 and info['generic_tiles'] == 0, info
            if name == 'falc_cah.npz':
                assert info['tiles'] == 66 and info['generic_tiles'] == 0 and info['linked_tiles'] > 0, info
            print(name, 'compact' if compact else 'ray-dependent', info)
    # the toy topologies of tests/test_toy_topologies.py (incl. multiplets under linked continua)
    shapes = {}
    for kw in CASES:
        prob, _ = toy_problem(**dict(kw, ncol=1))
        for bits in (0, 1, 8):
            rc, msg, s, tiles = lib.probe(prob, bits)
            assert rc == 0, (kw, bits, msg)
            n += 1
            if bits == 0 and kw.get('multiplet'):
                shapes[kw['multiplet']] = sorted(set((int(t[2]), int(t[4]), int(t[5] > 0), int(t[6])) for t in tiles))
    # three overlapping lines under linked continua run the <3, 3, linked> instance; four have no instance -> generic linked (-3)
    assert any(nP == 3 and nL == 3 and lk and code == 64 + 3 * 8 + 3 for nP, nL, lk, code in shapes[3]), shapes[3]
    assert any(nP == 4 and nL == 4 and lk and code == -3 for nP, nL, lk, code in shapes[4]), shapes[4]
    assert not any(code == 64 + 4 * 8 + 4 for *_, code in shapes[4])
    # random tables: every plan that is accepted satisfies the invariants; refusals are the documented ones
    rng = np.random.default_rng(2024)
    refused = {}
    for _ in range(400):
        prob = random_problem(rng)
        rc, msg, s, tiles = lib.probe(prob, int(rng.choice([0, 0, 1, 2, 4, 8, 16])))
        assert rc in (0, 1, 5), (rc, msg)          # ok, LSX_EINVAL, LSX_EUNSUPPORTED; -1000 = an invariant failed
        if rc:
            key = msg.split(':')[1].strip()[:36] if ':' in msg else msg
            refused[key] = refused.get(key, 0) + 1
        n += 1
    # degenerate descriptors are refused, not crashed on
    prob, _, _ = fixtures.load_problem_npz(golden('falc_ca.npz'))
    for edit in ('Nspace2', 'range', 'levels'):
        q, _, _ = fixtures.load_problem_npz(golden('falc_ca.npz'))
        if edit == 'Nspace2':
            q.Nspace = 2
        elif edit == 'range':
            q.trans[0].Nblue = q.Nspect - 3
        else:
            q.trans[0].j = 99
        rc, msg, s, tiles = lib.probe(q, 0)
        assert rc != 0 and msg, edit
    print('plan leg: %d plans verified; refusals among random tables: %s' % (n, refused))


def leg_oracle(path):
    from toy import toy_problem
    lib = _capi.LsxLibrary(path)
    assert lib.backend == 'oracle-c'
    from lightspinner_amd import Engine, synth
    from lightspinner_amd.rh_method import Context
    import grid_cases
    for name, iters in (('falc_ca.npz', 5), ('falc_cah.npz', 4), ('falc_ca_vlos.npz', 4)):
        prob, block, raw = fixtures.load_problem_npz(golden(name), phi_compact=(name != 'falc_ca_vlos.npz'))
        for solver in ('linear', 'parabolic'):
            e = Engine(prob, 1, lib=lib)
            e.set_columns(0, block)
            e.set_formal_solver(solver)
            for it in range(iters):
                e.formal_sol_gamma()
                if it >= 3:
                    e.stat_equil()
            assert np.isfinite(e.get(_capi.LSX_N)).all()
            e.close()
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'), phi_compact=False)
    blk, prof = synth.perturbed_columns(prob, base, raw, ncol=3, seed=5)
    e = Engine(prob, 3, lib=lib)
    synth.load_columns(e, blk, prof)            # lsx_set_line_profiles: the oracle's Voigt
    e.formal_sol_gamma(); e.stat_equil()
    # the look-ahead loop: speculative formal solutions (a copy of everything they overwrite), the discard at the end
    from lightspinner_amd import drivers
    h = drivers.iterate_mali_engine(e, max_iter=6, pipelined=True)
    assert h.n_iter == 6
    e.formal_sol_gamma_speculative(); e.discard_formal_sol()
    e.close()
    for kw in (dict(seed=1), dict(seed=3, Nrays=5, Nspace=82, Nspect=130, sca_per_lambda=True), dict(seed=15, Nrays=5, Nspace=41, Nspect=120, multiplet=4)):
        prob, block = toy_problem(**kw)
        e = Engine(prob, block.ncol, lib=lib)
        e.set_columns(0, block)
        for it in range(4):
            e.formal_sol_gamma()
            if it >= 2:
                e.stat_equil()
        e.close()
    # set-up chain + the drop-in Context on reference-shaped objects
    import context_cases
    context_cases.context_native_setup_chain(lib, 'falc_cah.npz')
    context_cases.piecewise_linear_1d_dropin(lib)
    grid_cases.reference_grid_bit_exact(lib)
    grid_cases.line_grids_and_continuum_alpha(lib)
    grid_cases.random_and_edge_cases(lib)
    print('oracle leg: done')


if __name__ == '__main__':
    {'plan': leg_plan, 'oracle': leg_oracle}[sys.argv[1]](sys.argv[2])
    print('SANITIZED RUN COMPLETE')
