"""Synthetic atmosphere columns for the multi-column configurations:
FALC-perturbed columns defined at the hot-path-input level.

Column 0 is the unperturbed base column.  Column c > 0 uses numpy.random.default_rng(seed + c):
smooth multiplicative log-normal factors (sigma, correlation length in depth points) on the
background opacity (and emissivity by the same factor), on the populations (nStar, n, nTotal
jointly), on the collisional rates, and -- when vlos_sigma > 0 -- a smooth line-of-sight velocity
so that the line profiles become genuinely ray dependent (built on the device by lsx_set_line_profiles from the base
column's damping parameters and broadening velocities)."""
import numpy as np

from .problem import ColumnBlock, Problem


def _smooth_field(rng, Ns, corr):
    x = rng.normal(size=Ns + 6 * corr)
    kern = np.exp(-0.5 * (np.arange(-3 * corr, 3 * corr + 1) / corr) ** 2)
    kern /= np.sqrt(np.sum(kern ** 2))
    return np.convolve(x, kern, mode='valid')[:Ns]


def perturbed_columns(prob: Problem, base: ColumnBlock, raw: dict, ncol: int, seed: int = 1234, sigma: float = 0.05,
                      corr: int = 8, vlos_sigma: float = 2.0e3, first: int = 0):
    """columns [first, first + ncol) of the synthetic ensemble (deterministic per absolute index)
    -> (ColumnBlock, profile inputs or None).

    With a line-of-sight velocity (vlos_sigma > 0 on a context that is not phi_compact) the line profiles are ray
    dependent and are NOT built here: the block has phi = wphi = None and the second value is (aDamp [ncol][Nlines][Nspace],
    vBroad [ncol][Natoms][Nspace], vlos [ncol][Nspace]) for Engine.set_line_profiles -- compute_phi then runs on the device
    (rh_method.py:198-243).  Without one the base column's profiles are copied and the second value is None."""
    Ns = prob.Nspace
    f = lambda a: np.repeat(np.asarray(a), ncol, axis=0).copy()
    out = {k: f(getattr(base, k)) for k in ('height', 'temperature', 'nStar', 'nTotal', 'n', 'C', 'bg_chi', 'bg_eta', 'bg_sca')}
    use_vlos = vlos_sigma > 0 and not prob.phi_compact and prob.Nlines > 0
    vlos = np.zeros((ncol, Ns))
    for q in range(ncol):
        c = first + q
        if c == 0:
            continue
        rng = np.random.default_rng(seed + c)
        fb = np.exp(sigma * _smooth_field(rng, Ns, corr))
        fn = np.exp(sigma * _smooth_field(rng, Ns, corr))
        fc = np.exp(sigma * _smooth_field(rng, Ns, corr))
        out['bg_chi'][q] *= fb
        out['bg_eta'][q] *= fb
        for k in ('nStar', 'n', 'nTotal'):
            out[k][q] *= fn
        out['C'][q] *= fc
        if use_vlos:
            vlos[q] = vlos_sigma * _smooth_field(rng, Ns, corr)
    if use_vlos:
        lines = [kr for kr, t in enumerate(prob.trans) if t.is_line]
        aD = np.repeat(np.stack([raw['t%d_aDamp' % kr] for kr in lines])[None], ncol, axis=0)
        vB = np.repeat(np.stack([raw['a%d_vBroad' % a] for a in range(prob.Natoms)])[None], ncol, axis=0)
        return ColumnBlock(phi=None, wphi=None, **out).validate(prob), (aD, vB, vlos)
    if base.phi is None:
        raise ValueError('the base column carries no profiles to copy')
    return ColumnBlock(phi=f(base.phi), wphi=f(base.wphi), **out).validate(prob), None


def load_columns(engine, block: ColumnBlock, prof=None, col0: int = 0, step: int = 100):
    """block (+ profile inputs) -> engine: lsx_set_columns in chunks, then lsx_set_line_profiles where the profiles were
    not handed over"""
    for a in range(0, block.ncol, step):
        engine.set_columns(col0 + a, block.slice(a, min(block.ncol, a + step)))
    if prof is not None:
        engine.set_line_profiles(col0, prof[0], prof[1], prof[2])


def perturbed_atmospheres(prob: Problem, raw: dict, ncol: int, seed: int = 4321, sigma_T: float = 0.02, sigma_n: float = 0.05,
                          corr: int = 8, vlos_sigma: float = 2.0e3, first: int = 0):
    """atmospheres for lsx_set_atmosphere, columns [first, first + ncol) of an ensemble around the fixture's atmosphere
    (`raw`: a problem file, which holds temperature, ne, vturb, hGround and the atoms' nTotal): smooth log-normal factors on the
    temperature (sigma_T), jointly on the electron and total densities (sigma_n), on the microturbulence, and a smooth
    line-of-sight velocity.  With lte_pops the library then derives broadening, damping, LTE populations, collisional rates
    and line profiles that are consistent with each column's own atmosphere -- unlike perturbed_columns,
    which perturbs the hot path's inputs directly.  Column 0 is the unperturbed atmosphere.
    -> dict(temperature, ne, vturb, nHGround [ncol][Nspace], nTotal [ncol][Natoms][Nspace], vlos or None)"""
    Ns = prob.Nspace
    rep = lambda a: np.repeat(np.asarray(a, dtype=np.float64)[None], ncol, axis=0).copy()
    out = dict(temperature=rep(raw['temperature']), ne=rep(raw['ne']), vturb=rep(raw['vturb']), nHGround=rep(raw['hGround']),
               nTotal=rep(np.stack([raw['a%d_nTotal' % a] for a in range(prob.Natoms)])))
    use_vlos = vlos_sigma > 0 and not prob.phi_compact
    vlos = np.zeros((ncol, Ns))
    for q in range(ncol):
        c = first + q
        if c == 0:
            continue
        rng = np.random.default_rng(seed + c)
        fT = np.exp(sigma_T * _smooth_field(rng, Ns, corr))
        fn = np.exp(sigma_n * _smooth_field(rng, Ns, corr))
        fv = np.exp(sigma_n * _smooth_field(rng, Ns, corr))
        out['temperature'][q] *= fT
        out['ne'][q] *= fn
        out['nHGround'][q] *= fn
        out['nTotal'][q] *= fn
        out['vturb'][q] *= fv
        if use_vlos:
            vlos[q] = vlos_sigma * _smooth_field(rng, Ns, corr)
    out['vlos'] = vlos if use_vlos else None
    return out
