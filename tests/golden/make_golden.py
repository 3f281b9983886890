#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the UNMODIFIED
reference from /root/reference (build container only; the reference never
travels to the GPU box -- only the numeric .npz files written here do).

The reference imports numba / astropy / specutils at module top; none is
installed in this image, so harness-side stand-ins from tests/golden/_refstubs
are put on sys.path (numba.njit == identity, i.e. the reference's pure-Python
semantics).  One numpy-2 compatibility patch is applied at run time, nothing
under /root/reference is touched:
  * atomic_model.avoid_recursion_eq     returns False for shape-mismatched
    arrays (numpy<1.25 semantics; only used for list membership in setup).
(np.bool, used at rh_method.py:124, exists again in numpy 2.x.)

Usage:  python tests/golden/make_golden.py [units] [falc_ca] [falc_cah] [falc_ca_vlos] [falc_c] [falc_fe] [falc_mg] [falc_all] [rf] [rf_inputs] [setup] [all]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('LIGHTSPINNER_REF', '/root/reference')
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, '_refstubs'))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402

import atomic_model  # noqa: E402

_orig_eq = atomic_model.avoid_recursion_eq


def _compat_eq(a, b):
    if isinstance(a, np.ndarray):
        if not isinstance(b, np.ndarray) or a.shape != b.shape:
            return False
    return _orig_eq(a, b)


atomic_model.avoid_recursion_eq = _compat_eq

import formal_solver  # noqa: E402
from fal import Falc82  # noqa: E402
from rh_atoms import CaII_atom, H_6_atom, C_atom, Fe_simple_atom, MgII_atom  # noqa: E402
from atomic_set import RadiativeSet  # noqa: E402
from rh_method import Context  # noqa: E402
from background import Background  # noqa: E402
from atomic_model import AtomicLine  # noqa: E402


# ----------------------------------------------------------------------------
def build_ctx(active, vlos=None, temp_pert=None, start_pops=None, nrays=5, models=None):
    """Reproduces the setup of test.py:8-18 / response_fn.py:23-37.  models: the RadiativeSet's atomic models (default: test.py's
    CaII + H); hydrogen must be among them (its ground-level populations enter the van der Waals damping, rh_method.py:223)."""
    ac = Falc82()
    ac.quadrature(nrays)
    if temp_pert is not None:
        k, dT = temp_pert
        ac.temperature[k] += dT  # Quantity in K; same as response_fn.py:26
    if vlos is not None:
        ac.vlos[:] = vlos  # m/s (constructor already converted to SI)
    atmos = ac.convert_scales()
    aSet = RadiativeSet(models if models is not None else [CaII_atom(), H_6_atom()])
    aSet.set_active(*active)
    spect = aSet.compute_wavelength_grid()
    eqPops = aSet.compute_eq_pops(atmos)
    if start_pops is not None:
        for name, p in start_pops.items():
            eqPops[name].pops = np.copy(p)
    background = Background(atmos, spect)
    ctx = Context(atmos, spect, eqPops, background)
    return ctx


def dump_inputs(ctx, full_phi=False, phi_sample_only=False):
    """Flat dict of the post-setup hot-path inputs."""
    atmos, spect, bg = ctx.atmos, ctx.spect, ctx.background
    d = {}
    d['wavelength'] = np.array(spect.wavelength)
    for k in ('muz', 'wmu', 'height', 'temperature', 'ne', 'vlos', 'vturb', 'nHTot'):
        d[k] = np.array(getattr(atmos, k), dtype=np.float64)
    d['bg_chi'] = np.array(bg.chi)
    d['bg_eta'] = np.array(bg.eta)
    sca = np.array(bg.sca)
    if np.all(sca == sca[0:1]):
        d['bg_sca'] = sca[0].copy()
    else:
        d['bg_sca'] = sca
    d['hGround'] = np.array(ctx.eqPops['H'].n[0])
    names = []
    t_atom, t_isline, t_i, t_j, t_nblue, t_nl = [], [], [], [], [], []
    t_A, t_Bji, t_Bij, t_l0 = [], [], [], []
    t_active = []
    kr = 0
    for a, atom in enumerate(ctx.activeAtoms):
        # compute_collisions is re-run at the top of every FS call; C is a pure
        # function of the atmosphere (rh_method.py:474-487), capture it here.
        atom.compute_collisions()
        names.append(atom.atomicModel.name)
        d['a%d_nStar' % a] = np.array(atom.nStar)
        d['a%d_nTotal' % a] = np.array(atom.nTotal)
        d['a%d_n0' % a] = np.array(atom.n)
        d['a%d_C' % a] = np.array(atom.C)
        d['a%d_vBroad' % a] = np.array(atom.vBroad)
        d['a%d_weight' % a] = np.float64(atom.atomicTable[atom.atomicModel.name].weight)
        for t in atom.trans:
            t_atom.append(a)
            t_isline.append(1 if t.isLine else 0)
            t_i.append(t.i)
            t_j.append(t.j)
            t_nblue.append(int(t.Nblue))
            t_nl.append(t.wavelength.shape[0])
            t_active.append(np.array(t.active, dtype=np.bool_))
            d['t%d_wavelength' % kr] = np.array(t.wavelength)
            if t.isLine:
                t_A.append(t.Aji); t_Bji.append(t.Bji); t_Bij.append(t.Bij); t_l0.append(t.lambda0)
                d['t%d_wphi' % kr] = np.array(t.wphi)
                aDamp, _ = t.transModel.damping(atmos, atom.vBroad, atom.hPops.n[0])
                d['t%d_aDamp' % kr] = np.array(aDamp)
                phi = np.array(t.phi)
                same = np.all(phi == phi[:, :1, :1, :])
                if phi_sample_only:
                    d['t%d_phi_sample' % kr] = phi[::7, :, :, ::9].copy()
                elif same and not full_phi:
                    d['t%d_phi' % kr] = phi[:, 0, 0, :].copy()
                else:
                    d['t%d_phi' % kr] = phi
            else:
                t_A.append(0.0); t_Bji.append(0.0); t_Bij.append(0.0); t_l0.append(0.0)
                d['t%d_alpha' % kr] = np.array(t.alpha)
            kr += 1
    d['atom_names'] = np.array(names)
    d['t_atom'] = np.array(t_atom, dtype=np.int32)
    d['t_isline'] = np.array(t_isline, dtype=np.int32)
    d['t_i'] = np.array(t_i, dtype=np.int32)
    d['t_j'] = np.array(t_j, dtype=np.int32)
    d['t_Nblue'] = np.array(t_nblue, dtype=np.int32)
    d['t_Nlambda'] = np.array(t_nl, dtype=np.int32)
    d['t_Aji'] = np.array(t_A); d['t_Bji'] = np.array(t_Bji)
    d['t_Bij'] = np.array(t_Bij); d['t_lambda0'] = np.array(t_l0)
    d['t_active'] = np.array(t_active)
    return d


def snap_fs(ctx, d, tag, dJ, with_rates=False):
    d['%s_J' % tag] = ctx.J.copy()
    d['%s_I' % tag] = ctx.I.copy()
    d['%s_dJ' % tag] = np.float64(dJ)
    for a, atom in enumerate(ctx.activeAtoms):
        d['%s_Gamma_a%d' % (tag, a)] = atom.Gamma.copy()
    if with_rates:
        kr = 0
        for atom in ctx.activeAtoms:
            for t in atom.trans:
                d['%s_Rij_t%d' % (tag, kr)] = t.Rij.copy()
                d['%s_Rji_t%d' % (tag, kr)] = t.Rji.copy()
                kr += 1


def snap_se(ctx, d, tag, dPops):
    d['%s_dPops' % tag] = np.float64(dPops)
    for a, atom in enumerate(ctx.activeAtoms):
        d['%s_n_a%d' % (tag, a)] = atom.n.copy()


def run_mali(ctx, d, snap_iters=(1, 2, 3, 4, 5), max_iter=500, stop_after=None, log=None):
    """The driver loop of test.py:20-29 with snapshots."""
    dJ, dPops, i = 1.0, 1.0, 0
    traj_dJ, traj_dP = [], []
    t0 = time.time()
    while dJ > 2e-3 or dPops > 1e-3:
        i += 1
        dJ = ctx.formal_sol_gamma_matrices()
        if i in snap_iters:
            snap_fs(ctx, d, 'fs%d' % i, dJ, with_rates=(i <= 2))
        if i > 3:
            dPops = ctx.stat_equil()
            if i in snap_iters:
                snap_se(ctx, d, 'se%d' % i, dPops)
        traj_dJ.append(dJ)
        traj_dP.append(dPops if i > 3 else np.nan)
        if log:
            print('%s it %03d dJ %.6e dPops %.6e  (%.1fs)' % (log, i, dJ, dPops, time.time() - t0), flush=True)
        if stop_after is not None and i >= stop_after:
            break
        if i >= max_iter:
            break
    d['traj_dJ'] = np.array(traj_dJ)
    d['traj_dPops'] = np.array(traj_dP)
    d['n_iter'] = np.int32(i)
    return i


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print('wrote %s (%.1f kB)' % (path, os.path.getsize(path) / 1e3), flush=True)


# ----------------------------------------------------------------------------
def gen_units():
    """w2 / piecewise_1d_impl / piecewise_linear_1d / planck unit vectors."""
    from utils import planck
    d = {}
    dt = np.concatenate([np.logspace(-12, 9, 400),
                         [5e-4, np.nextafter(5e-4, 0), np.nextafter(5e-4, 1),
                          50.0, np.nextafter(50.0, 0), np.nextafter(50.0, 100.0), 0.0]])
    d['w2_dtau'] = dt
    d['w2_out'] = np.array([formal_solver.w2(x) for x in dt])
    rng = np.random.default_rng(20260101)
    cases = []
    for c, N in enumerate([3, 4, 5, 17, 82, 82, 82, 83, 200, 41]):
        z = np.sort(rng.uniform(-1e5, 2e6, N))[::-1].copy()
        logchi = np.cumsum(rng.normal(0.15, 0.3, N)) + rng.uniform(-16, -9)
        chi = np.exp(logchi)
        if c == 5:
            chi *= 1e6  # mostly saturated branch
        if c == 6:
            chi *= 1e-6  # mostly Taylor branch
        S = np.exp(rng.normal(-20, 1.0, N)) * np.linspace(0.1, 3.0, N)
        mu = float(rng.uniform(0.04, 1.0))
        for toFrom in (False, True):
            Istart = float(rng.uniform(0, 1e-8)) if toFrom else 0.0
            I, Psi = formal_solver.piecewise_1d_impl(mu, toFrom, Istart, z, chi, S)
            tag = 'pw%d_%d' % (c, int(toFrom))
            d[tag + '_z'] = z; d[tag + '_chi'] = chi; d[tag + '_S'] = S
            d[tag + '_mu'] = np.float64(mu); d[tag + '_Istart'] = np.float64(Istart)
            d[tag + '_I'] = I; d[tag + '_Psi'] = Psi
        cases.append(N)
    d['pw_N'] = np.array(cases, dtype=np.int32)

    # piecewise_linear_1d with boundary conditions on a duck-typed atmosphere
    class A:
        pass
    at = A()
    at.Nspace = 82
    at.muz = np.array([0.2, 0.9])
    at.height = np.sort(rng.uniform(-1e5, 2e6, 82))[::-1].copy()
    at.temperature = np.linspace(4000.0, 9500.0, 82)
    chi = np.exp(np.cumsum(rng.normal(0.2, 0.3, 82)) - 14)
    S = np.exp(rng.normal(-19, 0.5, 82))
    d['pl_height'] = at.height; d['pl_temperature'] = at.temperature; d['pl_muz'] = at.muz
    d['pl_chi'] = chi; d['pl_S'] = S
    wavs = np.array([30.0, 393.4, 854.2, 2000.0])
    d['pl_wav'] = wavs
    for wi, wav in enumerate(wavs):
        for mu in range(2):
            for toFrom in (False, True):
                r = formal_solver.piecewise_linear_1d(at, mu, toFrom, wav, chi, S)
                d['pl_I_%d_%d_%d' % (wi, mu, int(toFrom))] = r.I
                d['pl_Psi_%d_%d_%d' % (wi, mu, int(toFrom))] = r.PsiStar
    T = np.array([3000.0, 4500.0, 6000.0, 9000.0, 1e5])
    d['planck_T'] = T
    d['planck_wav'] = wavs
    d['planck_B'] = np.array([planck(T, w) for w in wavs])
    save('units.npz', d)


def gen_falc_ca():
    ctx = build_ctx(['Ca'])
    d = dump_inputs(ctx)
    n = run_mali(ctx, d, log='falc_ca')
    snap_fs(ctx, d, 'conv', d['traj_dJ'][-1])
    snap_se(ctx, d, 'conv', d['traj_dPops'][-1])
    print('falc_ca converged in', n)
    save('falc_ca.npz', d)
    return ctx


def gen_falc_cah(stop_after=8):
    ctx = build_ctx(['Ca', 'H'])
    d = dump_inputs(ctx)
    run_mali(ctx, d, snap_iters=(1, 2, 4, 5), stop_after=stop_after, log='falc_cah')
    snap_fs(ctx, d, 'last', d['traj_dJ'][-1])
    snap_se(ctx, d, 'last', d['traj_dPops'][-1])
    # keep the fixture small: drop the bulky intermediate J snapshots except fs1/fs2/last
    for k in list(d.keys()):
        if k in ('fs4_J', 'fs5_J'):
            del d[k]
    save('falc_cah.npz', d)


def gen_falc_ca_vlos():
    """Non-zero line-of-sight velocity: phi is genuinely 4-D (rh_method.py:229-240).
    Only the ingredients of phi (aDamp, vBroad, vlos) plus a strided sample of the
    reference's phi are stored; tests rebuild phi with the same scipy wofz."""
    k = np.arange(82)
    vlos = 4.0e3 * np.sin(2 * np.pi * k / 41.0) * np.exp(-((k - 35.0) / 25.0) ** 2) + 1.5e3
    ctx = build_ctx(['Ca'], vlos=vlos)
    d = dump_inputs(ctx, phi_sample_only=True)
    run_mali(ctx, d, snap_iters=(1, 2, 4, 5), stop_after=6, log='falc_ca_vlos')
    snap_fs(ctx, d, 'last', d['traj_dJ'][-1])
    snap_se(ctx, d, 'last', d['traj_dPops'][-1])
    # the unperturbed-layout quantities identical to falc_ca are kept (file is self-contained)
    save('falc_ca_vlos.npz', d)


def gen_falc_multilevel(which):
    """FALC with one of the reference's larger model atoms active (rh_atoms.py:194 C_atom: 15 levels, 16 lines, 14 bound-free
    continua onto C II; :355 Fe_simple_atom: 15 levels, 15 lines, 14 continua; :50 MgII_atom: 11 levels, 15 lines, 10 continua):
    up to 14 transitions overlap at a wavelength, every continuum shares its atom with the lines it overlaps
    (rh_method.py:606-627, 654-681: the atom.eta / atom.chi / atom.U cross terms between lines and continua of one atom).
    Inputs + J, I, Gamma, dJ after formal solutions 1-4 and n, dPops after the first statistical equilibrium."""
    ctor, name = {'c': (C_atom, 'C'), 'fe': (Fe_simple_atom, 'Fe'), 'mg': (MgII_atom, 'MG')}[which]
    ctx = build_ctx([name], models=[H_6_atom(), ctor()])
    d = dump_inputs(ctx)
    run_mali(ctx, d, snap_iters=(1, 2, 3, 4), stop_after=4, log='falc_' + which)
    for k in ('fs2_J', 'fs3_J'):          # keep the file small: J after calls 1 and 4 (fs1_J is the J-dagger of call 2, and so on)
        del d[k]
    for k in list(d.keys()):              # the rates' snapshots pin I at every depth: keep them for call 1 only
        if k.startswith('fs2_R'):
            del d[k]
    save('falc_%s.npz' % which, d)


def gen_falc_all():
    """FALC with ALL FIVE of the reference's model atoms in the set and active at once (rh_atoms.py:4 H_6, :50 MgII, :152 CaII,
    :194 C, :355 Fe_simple): 53 levels, 109 transitions, up to 44 bound-free continua of five atoms overlap at a wavelength, five Gamma matrices and five
    statistical equilibria per iteration (rh_method.py:586-590 the loop over the active atoms at a wavelength, :710 over the atoms
    of a statistical equilibrium).  Inputs + I, Gamma, dJ after formal solutions 1 and 4, J after 4, n and dPops after the first
    statistical equilibrium."""
    ctx = build_ctx(['H', 'Ca', 'MG', 'C', 'Fe'], models=[H_6_atom(), CaII_atom(), MgII_atom(), C_atom(), Fe_simple_atom()])
    d = dump_inputs(ctx)
    run_mali(ctx, d, snap_iters=(1, 4), stop_after=4, log='falc_all')
    del d['fs1_J']
    for k in list(d.keys()):
        if k.startswith('fs1_R'):
            del d[k]
    save('falc_all.npz', d)


def gen_rf(ks=(20, 48, 70)):
    """response_fn.py:23-67 for a handful of depth indices: delta-encoded inputs
    (only the depth-k entries differ) + converged emergent I."""
    base = build_ctx(['Ca'])
    dbase = dump_inputs(base)
    tmp = {}
    run_mali(base, tmp, snap_iters=(), log='rf_base')
    out = {'ks': np.array(ks, dtype=np.int32), 'tempPert': np.float64(50.0)}
    out['base_I'] = base.I.copy()
    out['base_n'] = base.activeAtoms[0].n.copy()
    out['base_niter'] = tmp['n_iter']
    startPops = {'Ca': base.eqPops['Ca'].n}
    for k in ks:
        for sgn, tag in ((+1, 'p'), (-1, 'm')):
            ctx = build_ctx(['Ca'], temp_pert=(k, sgn * 25.0), start_pops=startPops)
            dp = dump_inputs(ctx)
            # delta encode: store every array that differs from base, verifying where it differs
            for key, v in dp.items():
                b = dbase[key]
                if v.dtype.kind in 'US' or v.dtype.kind in 'ib':
                    assert np.array_equal(v, b), key
                    continue
                if key.endswith('_n0'):
                    continue  # warm start = base_n
                if np.array_equal(v, b):
                    continue
                diff = (v != b)
                where_k = np.zeros_like(diff)
                where_k[..., k] = True
                if key == 'height':
                    # uniform shift only: |dz| must be bitwise identical
                    assert np.array_equal(np.diff(v), np.diff(b)), 'height diff changed'
                    continue
                assert not np.any(diff & ~where_k), 'non-local change in %s' % key
                out['k%d%s_%s' % (k, tag, key)] = v[..., k].copy()
            t2 = {}
            n = run_mali(ctx, t2, snap_iters=(), log='rf_k%d%s' % (k, tag))
            out['k%d%s_I' % (k, tag)] = ctx.I.copy()
            out['k%d%s_n' % (k, tag)] = ctx.activeAtoms[0].n.copy()
            out['k%d%s_niter' % (k, tag)] = np.int32(n)
            out['k%d%s_traj_dJ' % (k, tag)] = t2['traj_dJ']
            out['k%d%s_traj_dPops' % (k, tag)] = t2['traj_dPops']
    save('rf_ca.npz', out)


def gen_rf_inputs(out_name='rf_ca_inputs.npz'):
    """response_fn.py:23-57, the INPUTS of all 2 x Nspace perturbed runs (T[k] +- 25 K), delta-encoded against the
    unperturbed column like gen_rf (only depth-k entries differ).  No converged reference outputs: running the
    reference's 164 MALI loops is hours of pure Python; parity of the response function is pinned on gen_rf's three
    depths, this file feeds the whole-workload run (bench.py --workload c5)."""
    base = build_ctx(['Ca'])
    dbase = dump_inputs(base)
    Ns = base.atmos.Nspace
    out = {'tempPert': np.float64(50.0), 'Nspace': np.int32(Ns)}
    t0 = time.time()
    for k in range(Ns):
        for sgn, tag in ((+1, 'p'), (-1, 'm')):
            ctx = build_ctx(['Ca'], temp_pert=(k, sgn * 25.0))
            dp = dump_inputs(ctx)
            for key, v in dp.items():
                b = dbase[key]
                if v.dtype.kind in 'USib':
                    assert np.array_equal(v, b), key
                    continue
                if key.endswith('_n0') or np.array_equal(v, b):
                    continue
                if key == 'height':
                    assert np.array_equal(np.diff(v), np.diff(b)), 'height diff changed'
                    continue
                diff = (v != b)
                where_k = np.zeros_like(diff)
                where_k[..., k] = True
                assert not np.any(diff & ~where_k), 'non-local change in %s' % key
                out['k%d%s_%s' % (k, tag, key)] = v[..., k].copy()
        print('rf inputs: depth %d / %d  (%.0f s)' % (k + 1, Ns, time.time() - t0), flush=True)
    save(out_name, out)


def _atom_data(model, d, pre, fresh):
    """numeric content of one AtomicModel (rh_atoms.py data, atomic_model.py derived constants) as flat arrays.
    `fresh` is an identical model on which compute_wavelength_grid has NOT run: its transitions still carry the local
    wavelength grids the merge starts from (atomic_model.py:347-380, 585-597, 645-660)."""
    import collisional_rates as cr
    from atomic_model import VdwUnsold, ExplicitContinuum, HydrogenicContinuum
    tab = model.atomicTable
    d[pre + 'weight'] = np.float64(tab[model.name].weight)
    d[pre + 'abundance'] = np.float64(tab[model.name].abundance)
    d[pre + 'lev_E_SI'] = np.array([l.E_SI for l in model.levels])
    d[pre + 'lev_g'] = np.array([l.g for l in model.levels])
    d[pre + 'lev_stage'] = np.array([l.stage for l in model.levels], dtype=np.int32)
    L = model.lines
    d[pre + 'line_i'] = np.array([l.i for l in L], dtype=np.int32)
    d[pre + 'line_j'] = np.array([l.j for l in L], dtype=np.int32)
    for key in ('f', 'gRad', 'stark', 'lambda0', 'Aji', 'Bji', 'Bij', 'qCore', 'qWing'):
        d[pre + 'line_' + key] = np.array([getattr(l, key) for l in L], dtype=np.float64)
    d[pre + 'line_NlambdaGen'] = np.array([l.NlambdaGen for l in L], dtype=np.int32)
    d[pre + 'line_vdw_unsold'] = np.array([1 if isinstance(l.vdw, VdwUnsold) else 0 for l in L], dtype=np.int32)
    d[pre + 'line_vdw_vals'] = np.array([list(l.vdw.vals)[:2] for l in L], dtype=np.float64).reshape(len(L), 2)
    d[pre + 'line_vdw_cross'] = np.array([getattr(l.vdw, 'cross', 0.0) for l in L], dtype=np.float64)
    for q, l in enumerate(fresh.lines):
        d[pre + 'line%d_grid0' % q] = np.array(l.wavelength)
    Cn = model.continua
    d[pre + 'cont_i'] = np.array([c.i for c in Cn], dtype=np.int32)
    d[pre + 'cont_j'] = np.array([c.j for c in Cn], dtype=np.int32)
    d[pre + 'cont_edge'] = np.array([c.lambdaEdge for c in Cn])
    d[pre + 'cont_minLambda'] = np.array([c.minLambda for c in Cn])
    d[pre + 'cont_hydrogenic'] = np.array([1 if isinstance(c, HydrogenicContinuum) else 0 for c in Cn], dtype=np.int32)
    d[pre + 'cont_alpha0'] = np.array([getattr(c, 'alpha0', 0.0) for c in Cn])
    for q, c in enumerate(fresh.continua):
        d[pre + 'cont%d_grid0' % q] = np.array(c.wavelength)
        d[pre + 'cont%d_alpha_grid0' % q] = np.array(c.alpha)
    for q, c in enumerate(Cn):
        d[pre + 'cont%d_alpha' % q] = np.array(c.alpha)             # on the merged grid (compute_alpha)
        d[pre + 'cont%d_wavelength' % q] = np.array(c.wavelength)
    kinds = {cr.Omega: 0, cr.CI: 1, cr.CE: 2}
    K = model.collisions
    d[pre + 'col_kind'] = np.array([kinds[type(c)] for c in K], dtype=np.int32)
    d[pre + 'col_i'] = np.array([c.i for c in K], dtype=np.int32)
    d[pre + 'col_j'] = np.array([c.j for c in K], dtype=np.int32)
    nt = max(len(c.temperature) for c in K)
    d[pre + 'col_nT'] = np.array([len(c.temperature) for c in K], dtype=np.int32)
    T = np.zeros((len(K), nt)); R = np.zeros((len(K), nt))
    for q, c in enumerate(K):
        T[q, :len(c.temperature)] = c.temperature
        R[q, :len(c.rates)] = c.rates
    d[pre + 'col_T'] = T
    d[pre + 'col_rates'] = R


def gen_setup():
    """Set-up chain that defines the hot-path inputs: atomic data of the two models of
    test.py, the wavelength-grid merge (atomic_set.py:377-455) and, for FALC and for a perturbed atmosphere, the
    reference's v_broad (atomic_model.py:66-69), damping (:491-502: radiative + Unsold van der Waals :166-198 + Stark
    :300-345), collisional rates (collisional_rates.py:10-96 with scipy's cubic interp1d) and LTE populations
    (atomic_set.py:105-145)."""
    d = {}
    variants = []
    k = np.arange(82)
    for tag, dT, fv, fne in (('atm0', None, 1.0, 1.0),
                             ('atm1', 600.0 * np.sin(2 * np.pi * k / 30.0) * np.exp(-((k - 40.0) / 30.0) ** 2) + 150.0, 1.7, 0.6)):
        ac = Falc82()
        ac.quadrature(5)
        if dT is not None:
            ac.temperature[:] = ac.temperature + dT * ac.temperature.unit
            ac.vturb[:] = ac.vturb * fv
            ac.ne[:] = ac.ne * fne
        atmos = ac.convert_scales()
        aSet = RadiativeSet([CaII_atom(), H_6_atom()])
        aSet.set_active('Ca', 'H')
        fresh = {m.name: m for m in RadiativeSet([CaII_atom(), H_6_atom()]).atoms}
        spect = aSet.compute_wavelength_grid()
        eqPops = aSet.compute_eq_pops(atmos)
        background = Background(atmos, spect)
        ctx = Context(atmos, spect, eqPops, background)
        for key in ('temperature', 'ne', 'vturb', 'nHTot'):
            d['%s_%s' % (tag, key)] = np.array(getattr(atmos, key), dtype=np.float64)
        d['%s_hGround' % tag] = np.array(eqPops['H'].n[0])
        for a, atom in enumerate(ctx.activeAtoms):
            pre = '%s_a%d_' % (tag, a)
            atom.compute_collisions()
            d[pre + 'vBroad'] = np.array(atom.vBroad)
            d[pre + 'nStar'] = np.array(atom.nStar)
            d[pre + 'nTotal'] = np.array(atom.nTotal)
            d[pre + 'C'] = np.array(atom.C)
            lines = [t for t in atom.trans if t.isLine]
            aD, Qe = [], []
            for t in lines:
                a_, q_ = t.transModel.damping(atmos, atom.vBroad, atom.hPops.n[0])
                aD.append(a_); Qe.append(q_)
            d[pre + 'aDamp'] = np.array(aD)
            d[pre + 'Qelast'] = np.array(Qe)
        if tag == 'atm0':
            d['atom_names'] = np.array([a.atomicModel.name for a in ctx.activeAtoms])
            tabH, tabHe = ctx.activeAtoms[0].atomicTable['H'], ctx.activeAtoms[0].atomicTable['He']
            d['weight_H'] = np.float64(tabH.weight); d['weight_He'] = np.float64(tabHe.weight)
            d['abundance_He'] = np.float64(tabHe.abundance)
            for a, atom in enumerate(ctx.activeAtoms):
                _atom_data(atom.atomicModel, d, 'm%d_' % a, fresh[atom.atomicModel.name])
            # the merge itself: inputs are the *_grid0 arrays + continuum edges + lambdaReference; outputs:
            d['grid_lambdaReference'] = np.float64(500.0)
            d['grid_wavelength'] = np.array(spect.wavelength)
            # spect.transitions order = iteration order of a Python set of models: record which (atom, kind, index) each is
            order = []
            for t in spect.transitions:
                a = [x.atomicModel for x in ctx.activeAtoms].index(t.atom)
                m = ctx.activeAtoms[a].atomicModel
                isline = isinstance(t, AtomicLine)
                q = [id(x) for x in (m.lines if isline else m.continua)].index(id(t))
                order.append((a, 1 if isline else 0, q))
            d['grid_trans_order'] = np.array(order, dtype=np.int32)
            d['grid_blueIdx'] = np.array([int(b) for b in spect.blueIdx], dtype=np.int32)
            d['grid_Nlambda'] = np.array([t.wavelength.shape[0] for t in spect.transitions], dtype=np.int32)
            act = np.zeros((len(spect.transitions), spect.wavelength.shape[0]), dtype=np.bool_)
            ids = [id(t) for t in spect.transitions]
            for la, sset in enumerate(spect.activeSet):
                for t in sset:
                    act[ids.index(id(t)), la] = True
            d['grid_active'] = act
    save('setup_falc.npz', d)


if __name__ == '__main__':
    what = sys.argv[1:] or ['all']
    if 'all' in what:
        what = ['units', 'falc_ca', 'falc_cah', 'falc_ca_vlos', 'falc_c', 'falc_fe', 'falc_mg', 'falc_all', 'rf']
    for w in what:
        t0 = time.time()
        {'units': gen_units, 'falc_ca': gen_falc_ca, 'falc_cah': gen_falc_cah,
         'falc_ca_vlos': gen_falc_ca_vlos, 'rf': gen_rf, 'falc_c': lambda: gen_falc_multilevel('c'),
         'falc_fe': lambda: gen_falc_multilevel('fe'), 'falc_mg': lambda: gen_falc_multilevel('mg'), 'falc_all': gen_falc_all, 'rf_inputs': gen_rf_inputs, 'setup': gen_setup}[w]()
        print('%s done in %.1fs' % (w, time.time() - t0), flush=True)
