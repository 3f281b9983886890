"""Read the flat "problem file" format (.npz) that tests/golden/make_golden.py writes
and bench.py consumes: post-setup hot-path inputs of one atmosphere column."""
import numpy as np

from .problem import Problem, Transition, ColumnBlock
from . import lineprofile


def load_problem_npz(path_or_dict, phi_compact=None, rebuild_phi=False):
    """-> (Problem, ColumnBlock with ncol == 1, raw dict).

    phi_compact: None = use the compact [Nl][Nspace] form iff the file stores it;
    False = always expand to the full [Nl][Nrays][2][Nspace] layout of rh_method.py:224.
    rebuild_phi: recompute phi/wphi from (aDamp, vBroad, vlos) with the package's own
    compute_phi (files written with phi_sample_only carry no full phi)."""
    d = path_or_dict if isinstance(path_or_dict, dict) else dict(np.load(path_or_dict))
    wavelength = d['wavelength']
    Ns = d['height'].shape[0]
    Nrays = d['muz'].shape[0]
    names = [str(x) for x in d['atom_names']]
    Natoms = len(names)
    Nlevel = [d['a%d_nStar' % a].shape[0] for a in range(Natoms)]
    trans = []
    Ntrans = d['t_atom'].shape[0]
    have_full = True
    for kr in range(Ntrans):
        isline = bool(d['t_isline'][kr])
        t = Transition(atom=int(d['t_atom'][kr]), is_line=isline, i=int(d['t_i'][kr]), j=int(d['t_j'][kr]),
                       Nblue=int(d['t_Nblue'][kr]), Nlambda=int(d['t_Nlambda'][kr]),
                       Aji=float(d['t_Aji'][kr]), Bji=float(d['t_Bji'][kr]), Bij=float(d['t_Bij'][kr]),
                       lambda0=float(d['t_lambda0'][kr]))
        if not isline:
            t.alpha = d['t%d_alpha' % kr]
        elif ('t%d_phi' % kr) not in d:
            have_full = False
        trans.append(t)
    if not have_full:
        rebuild_phi = True

    phis, wphis = [], []
    stored_compact = True
    for kr, t in enumerate(trans):
        if not t.is_line:
            continue
        if rebuild_phi:
            ph, wp = lineprofile.compute_phi(d['t%d_wavelength' % kr], t.lambda0, d['t%d_aDamp' % kr],
                                             d['a%d_vBroad' % t.atom], d['vlos'], d['muz'], d['wmu'])
            stored_compact = False
        else:
            ph, wp = d['t%d_phi' % kr], d['t%d_wphi' % kr]
            if ph.ndim == 4:
                stored_compact = False
        phis.append(ph)
        wphis.append(wp)
    if phi_compact is None:
        phi_compact = stored_compact and all(p.ndim == 2 for p in phis)
    if phi_compact:
        if any(p.ndim != 2 for p in phis):
            raise ValueError('file holds a ray-dependent profile; phi_compact is not possible')
        phi = np.concatenate(phis, axis=0) if phis else np.zeros((0, Ns))
    else:
        full = [p if p.ndim == 4 else np.broadcast_to(p[:, None, None, :], (p.shape[0], Nrays, 2, Ns)) for p in phis]
        phi = np.concatenate(full, axis=0) if full else np.zeros((0, Nrays, 2, Ns))
    wphi = np.stack(wphis) if wphis else np.zeros((0, Ns))

    sca = d['bg_sca']
    prob = Problem(Nspace=Ns, wavelength=wavelength, muz=d['muz'], wmu=d['wmu'], Nlevel=Nlevel, trans=trans,
                   active=d['t_active'], sca_per_lambda=(sca.ndim == 2), phi_compact=bool(phi_compact),
                   atom_names=names)
    cat = lambda key: np.concatenate([d['a%d_%s' % (a, key)].reshape(-1, Ns) for a in range(Natoms)], axis=0)
    block = ColumnBlock(height=d['height'][None], temperature=d['temperature'][None],
                        nStar=cat('nStar')[None], nTotal=np.stack([d['a%d_nTotal' % a] for a in range(Natoms)])[None],
                        n=cat('n0')[None], C=cat('C')[None], bg_chi=d['bg_chi'][None], bg_eta=d['bg_eta'][None],
                        bg_sca=sca[None], phi=np.ascontiguousarray(phi)[None], wphi=wphi[None]).validate(prob)
    return prob, block, d


def gamma_from_raw(d, tag, prob):
    """concatenate the per-atom Gamma snapshots of a golden file into LSX_GAMMA layout"""
    return np.concatenate([d['%s_Gamma_a%d' % (tag, a)].reshape(-1, prob.Nspace) for a in range(prob.Natoms)], axis=0)


def pops_from_raw(d, tag, prob):
    return np.concatenate([d['%s_n_a%d' % (tag, a)] for a in range(prob.Natoms)], axis=0)
