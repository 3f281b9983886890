"""Small made-up model atoms with awkward level topologies, for HIP-vs-oracle parity on shapes the FALC
fixtures do not reach (the oracle itself is pinned on the reference's own atoms, tests/test_oracle_golden.py):

  * lines sharing a lower level and overlapping in wavelength,
  * a continuum whose lower level is another continuum's upper level (chained ionisation stages), so the
    fast-continuum epilogue has to take its generic level-cell path,
  * an atom with continua only,
  * odd Nspace, 1 / 3 / 5 rays (64 / 21 / 12 wavelengths per wavefront), per-wavelength scattering.

Nothing here comes from the reference; the numbers only need to give a well-posed transfer problem.
"""
import numpy as np

from lightspinner_amd.problem import Problem, Transition, ColumnBlock

HC = 6.62607004e-34 * 2.99792458e8
KB = 1.38064852e-23


def _gl(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * x + 0.5, 0.5 * w


def toy_problem(seed=0, Nspace=37, Nrays=3, Nspect=90, ncol=3, sca_per_lambda=False, phi_compact=False, chain=True,
                dead_level=False, multiplet=0):
    """dead_level: atom 1 gets a fourth level that no radiative transition touches and no collision populates
    (only collisions out of it): statistical equilibrium drives it to exactly 0, so the next stat_equil sees
    0/0 = NaN in its relative change (the case rh_method.py:741's builtin max drops)."""
    # atom 0: 5 levels.  lines 0-1 and 0-2 overlap; continua 1->4, 2->4 (simple set), and with chain=True a
    # "continuum" 0->1 whose upper level is the lower level of 1->4, and 3->4 that overlaps the lines
    specs0 = [('l', 0, 1, 0.30, 0.52), ('l', 0, 2, 0.45, 0.70), ('l', 1, 3, 0.80, 0.93),
              ('c', 1, 4, 0.00, 0.40), ('c', 2, 4, 0.05, 0.62), ('c', 3, 4, 0.20, 0.99)]
    if chain:
        specs0.append(('c', 0, 1, 0.00, 0.28))
    if multiplet:
        # a multiplet: `multiplet` lines from the ground level to distinct upper levels, all overlapping in wavelength, under
        # two bound-free continua of the same atom whose upper level (5) no line touches -- three / four per-ray slots in one
        # tile WITH linked continua (one of them starting on a line's upper level)
        assert 2 <= multiplet <= 4
        specs0 = [('l', 0, u, 0.30 + 0.02 * u, 0.60 + 0.03 * u) for u in range(1, multiplet + 1)]
        specs0 += [('c', 0, 5, 0.00, 0.95), ('c', 1, 5, 0.00, 0.55)]
    # atom 1: continua only, 3 levels
    specs1 = [('c', 0, 2, 0.00, 0.35), ('c', 1, 2, 0.10, 0.75)]
    Nlevel = [6 if multiplet else 5, 4 if dead_level else 3]
    return spec_problem([(Nlevel[0], specs0), (Nlevel[1], specs1)], seed=seed, Nspace=Nspace, Nrays=Nrays, Nspect=Nspect, ncol=ncol,
                        sca_per_lambda=sca_per_lambda, phi_compact=phi_compact, dead_atom=1 if dead_level else None)


def spec_problem(atoms, seed=0, Nspace=37, Nrays=5, Nspect=240, ncol=3, sca_per_lambda=False, phi_compact=False, dead_atom=None):
    """atoms: [(Nlevel, [(kind 'l' | 'c', lower level, upper level, blue end, red end as fractions of the spectrum), ...]), ...]:
    any level topology -- which transitions overlap in wavelength and which levels they share decides the tile classes the plan
    makes (lsx_plan.cpp), i.e. which template instances of the sweep kernels a problem reaches (tests/instance_cases.py)."""
    rng = np.random.default_rng(seed)
    wavelength = np.sort(rng.uniform(90.0, 900.0, Nspect))
    wavelength[1:] += np.arange(1, Nspect) * 1e-3          # strictly increasing
    muz, wmu = _gl(Nrays)

    def rng_range(lo_frac, hi_frac):
        a = int(lo_frac * Nspect)
        b = max(a + 3, int(hi_frac * Nspect))
        return a, min(b, Nspect) - a

    trans = []
    Nlevel = [int(nl) for nl, _ in atoms]
    dead_level = dead_atom is not None
    for atom, (_, specs) in enumerate(atoms):
        for kind, i, j, lo, hi in specs:
            Nblue, Nlam = rng_range(lo, hi)
            if kind == 'l':
                lam0 = wavelength[Nblue + Nlam // 2]
                Bij = rng.uniform(0.5, 2.0) * 1e9
                gij = rng.uniform(0.3, 1.5)
                Bji = gij * Bij
                Aji = 2.0 * HC / (lam0 * 1e-9) ** 3 * Bji
                trans.append(Transition(atom, True, i, j, Nblue, Nlam, Aji=Aji, Bji=Bji, Bij=Bij, lambda0=lam0))
            else:
                lam = wavelength[Nblue:Nblue + Nlam]
                alpha = rng.uniform(0.5, 2.0) * 1e-22 * (lam / lam[-1]) ** 3
                trans.append(Transition(atom, False, i, j, Nblue, Nlam, lambda0=lam[-1], alpha=alpha))
    Ntrans = len(trans)
    active = np.zeros((Ntrans, Nspect), dtype=np.uint8)
    for t, tr in enumerate(trans):
        active[t, tr.Nblue:tr.Nblue + tr.Nlambda] = 1
    prob = Problem(Nspace=Nspace, wavelength=wavelength, muz=muz, wmu=wmu, Nlevel=Nlevel, trans=trans, active=active,
                   sca_per_lambda=sca_per_lambda, phi_compact=phi_compact, atom_names=['X', 'Y', 'Z', 'W'][:len(Nlevel)])

    Ns = Nspace
    NLtot, NL2tot = prob.NLtot, prob.NL2tot
    depth = np.linspace(0.0, 1.0, Ns)
    cols = []
    for c in range(ncol):
        height = 2.0e6 * (1.0 - depth) ** 1.3 + np.sort(rng.uniform(0, 1e3, Ns))[::-1]
        height = np.sort(height)[::-1].copy()
        height += np.linspace(Ns, 0, Ns)                    # strictly decreasing
        temperature = 4500.0 + 5000.0 * depth ** 2 + 3000.0 * (1 - depth) ** 8 + rng.uniform(-50, 50, Ns)
        ntot = 1e14 * np.exp(9.0 * depth)                   # m^-3
        nStar = np.zeros((NLtot, Ns))
        nTotal = np.zeros((len(Nlevel), Ns))
        o = 0
        for a, nl in enumerate(Nlevel):
            frac = np.array([10.0 ** (-1.2 * l) for l in range(nl)])[:, None] * (1.0 + 0.3 * rng.uniform(-1, 1, (nl, Ns)))
            frac /= frac.sum(0)
            nTotal[a] = ntot * (1.0 if a == 0 else 0.3)
            nStar[o:o + nl] = frac * nTotal[a]
            o += nl
        n = nStar * (1.0 + 0.2 * rng.uniform(-1, 1, nStar.shape))
        o = 0
        for a, nl in enumerate(Nlevel):                     # keep sum(n) = nTotal like the reference's start
            n[o:o + nl] *= nTotal[a] / n[o:o + nl].sum(0)
            o += nl
        C = np.zeros((NL2tot, Ns))
        o = 0
        for a, nl in enumerate(Nlevel):
            Ca = 10.0 ** rng.uniform(1.0, 4.0, (nl, nl, Ns)) * (ntot / ntot[-1]) ** 0.5
            for l in range(nl):
                Ca[l, l] = 0.0
            if dead_level and a == dead_atom:
                Ca[nl - 1, :] = 0.0                         # C[to][from]: nothing goes INTO the last level
            C[o:o + nl * nl] = Ca.reshape(nl * nl, Ns)
            o += nl * nl
        bg_chi = 1e-9 * np.exp(11.0 * depth)[None, :] * rng.uniform(0.5, 2.0, (Nspect, 1)) * (1 + 0.1 * rng.uniform(-1, 1, (Nspect, Ns)))
        planck = 2.0 * HC * 2.99792458e8 / (wavelength[:, None] * 1e-9) ** 5 / np.expm1(HC / (wavelength[:, None] * 1e-9 * KB * temperature[None, :]))
        planck *= (wavelength[:, None] * 1e-9) ** 2 / 2.99792458e8          # per Hz
        bg_eta = bg_chi * planck * rng.uniform(0.7, 1.0, (Nspect, Ns))
        if sca_per_lambda:
            bg_sca = bg_chi * rng.uniform(0.05, 0.4, (Nspect, Ns))
        else:
            bg_sca = bg_chi.min(0) * rng.uniform(0.05, 0.4, Ns)
        SNl = prob.SNl
        if phi_compact:
            phi = np.zeros((SNl, Ns))
        else:
            phi = np.zeros((SNl, Nrays, 2, Ns))
        wphi = np.zeros((prob.Nlines, Ns))
        o = 0
        for li, tr in enumerate(prob.lines):
            lam = wavelength[tr.Nblue:tr.Nblue + tr.Nlambda]
            width = 0.15 * (lam[-1] - lam[0]) * (1.0 + 0.5 * depth)
            x = (lam[:, None] - tr.lambda0) / width[None, :]
            base = np.exp(-x * x) + 0.02 / (1.0 + x * x)
            if phi_compact:
                phi[o:o + tr.Nlambda] = base * 1e-7
            else:
                for m in range(Nrays):
                    for d in range(2):
                        shift = (1 if d else -1) * muz[m] * 0.1 * np.sin(3.0 * depth + c)
                        xs = x - shift[None, :]
                        phi[o:o + tr.Nlambda, m, d] = (np.exp(-xs * xs) + 0.02 / (1.0 + xs * xs)) * 1e-7
            wphi[li] = 1.0 / (base.sum(0) * 1e-7 * 1e6)
            o += tr.Nlambda
        cols.append(ColumnBlock(height=height[None], temperature=temperature[None], nStar=nStar[None], nTotal=nTotal[None],
                                n=n[None], C=C[None], bg_chi=bg_chi[None], bg_eta=bg_eta[None], bg_sca=bg_sca[None],
                                phi=phi[None], wphi=wphi[None]))
    block = ColumnBlock.concatenate(cols).validate(prob)
    return prob, block
