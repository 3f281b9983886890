"""Synthetic atmosphere columns for the multi-column configurations:
FALC-perturbed columns defined at the hot-path-input level.

Column 0 is the unperturbed base column.  Column c > 0 uses numpy.random.default_rng(seed + c):
smooth multiplicative log-normal factors (sigma, correlation length in depth points) on the
background opacity (and emissivity by the same factor), on the populations (nStar, n, nTotal
jointly), on the collisional rates, and -- when vlos_sigma > 0 -- a smooth line-of-sight velocity
so that the line profiles become genuinely ray dependent (rebuilt with lineprofile.compute_phi
from the base column's damping parameters and broadening velocities)."""
import numpy as np

from .problem import ColumnBlock, Problem
from . import lineprofile


def _smooth_field(rng, Ns, corr):
    x = rng.normal(size=Ns + 6 * corr)
    kern = np.exp(-0.5 * (np.arange(-3 * corr, 3 * corr + 1) / corr) ** 2)
    kern /= np.sqrt(np.sum(kern ** 2))
    return np.convolve(x, kern, mode='valid')[:Ns]


def perturbed_columns(prob: Problem, base: ColumnBlock, raw: dict, ncol: int, seed: int = 1234, sigma: float = 0.05,
                      corr: int = 8, vlos_sigma: float = 2.0e3, first: int = 0, device_profiles: bool = False):
    """columns [first, first + ncol) of the synthetic ensemble (deterministic per absolute index).

    device_profiles=True: the line profiles are not built here; returns (block with phi = wphi = None,
    (aDamp [ncol][Nlines][Nspace], vBroad [ncol][Natoms][Nspace], vlos [ncol][Nspace] or None)) for
    Engine.set_line_profiles -- compute_phi then runs on the device (rh_method.py:198-243)."""
    Ns = prob.Nspace
    f = lambda a: np.repeat(np.asarray(a), ncol, axis=0).copy()
    out = {k: f(getattr(base, k)) for k in ('height', 'temperature', 'nStar', 'nTotal', 'n', 'C', 'bg_chi', 'bg_eta',
                                            'bg_sca', 'wphi')}
    use_vlos = vlos_sigma > 0 and not prob.phi_compact and prob.Nlines > 0
    phi = None if device_profiles else np.empty((ncol,) + prob.phi_shape())
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
    if device_profiles:
        lines = [kr for kr, t in enumerate(prob.trans) if t.is_line]
        aD = np.repeat(np.stack([raw['t%d_aDamp' % kr] for kr in lines])[None], ncol, axis=0)
        vB = np.repeat(np.stack([raw['a%d_vBroad' % a] for a in range(prob.Natoms)])[None], ncol, axis=0)
        out['wphi'] = None
        return ColumnBlock(phi=None, **out).validate(prob), (aD, vB, vlos if use_vlos else None)
    if use_vlos:
        o, li = 0, 0
        for kr, t in enumerate(prob.trans):
            if not t.is_line:
                continue
            ph, wp = lineprofile.compute_phi(raw['t%d_wavelength' % kr], t.lambda0, raw['t%d_aDamp' % kr][None],
                                             raw['a%d_vBroad' % t.atom][None], vlos, prob.muz, prob.wmu)
            phi[:, o:o + t.Nlambda] = ph
            out['wphi'][:, li] = wp
            o += t.Nlambda
            li += 1
        if first == 0:  # column 0 stays bit-identical to the base column
            phi[0] = base.phi[0]
            out['wphi'][0] = base.wphi[0]
    else:
        phi[:] = base.phi
    return ColumnBlock(phi=phi, **out).validate(prob)
