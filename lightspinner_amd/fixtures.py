"""Read the flat "problem file" format (.npz) that tests/golden/make_golden.py writes
and bench.py consumes: post-setup hot-path inputs of one atmosphere column."""
import numpy as np

from .problem import Problem, Transition, ColumnBlock


def load_problem_npz(path_or_dict, phi_compact=None):
    """-> (Problem, ColumnBlock with ncol == 1, raw dict).

    phi_compact: None = use the compact [Nl][Nspace] form iff the file stores it;
    False = always expand to the full [Nl][Nrays][2][Nspace] layout of rh_method.py:224.
    Files that carry no full profile (written with phi_sample_only: vlos != 0) give a block with phi = wphi = None: the
    profiles are then built by the library from `profile_inputs(prob, raw)` (Engine.set_line_profiles)."""
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
    phi = wphi = None
    if have_full:
        lines = [kr for kr, t in enumerate(trans) if t.is_line]
        phis = [d['t%d_phi' % kr] for kr in lines]
        stored_compact = all(p.ndim == 2 for p in phis)
        if phi_compact is None:
            phi_compact = stored_compact
        if phi_compact:
            if not stored_compact:
                raise ValueError('file holds a ray-dependent profile; phi_compact is not possible')
            phi = np.concatenate(phis, axis=0) if phis else np.zeros((0, Ns))
        else:
            full = [p if p.ndim == 4 else np.broadcast_to(p[:, None, None, :], (p.shape[0], Nrays, 2, Ns)) for p in phis]
            phi = np.concatenate(full, axis=0) if full else np.zeros((0, Nrays, 2, Ns))
        phi = np.ascontiguousarray(phi)[None]
        wphi = (np.stack([d['t%d_wphi' % kr] for kr in lines]) if lines else np.zeros((0, Ns)))[None]
    else:
        if phi_compact:
            raise ValueError('file holds no full profile and a line-of-sight velocity; phi_compact is not possible')
        phi_compact = False

    sca = d['bg_sca']
    prob = Problem(Nspace=Ns, wavelength=wavelength, muz=d['muz'], wmu=d['wmu'], Nlevel=Nlevel, trans=trans,
                   active=d['t_active'], sca_per_lambda=(sca.ndim == 2), phi_compact=bool(phi_compact),
                   atom_names=names)
    cat = lambda key: np.concatenate([d['a%d_%s' % (a, key)].reshape(-1, Ns) for a in range(Natoms)], axis=0)
    block = ColumnBlock(height=d['height'][None], temperature=d['temperature'][None],
                        nStar=cat('nStar')[None], nTotal=np.stack([d['a%d_nTotal' % a] for a in range(Natoms)])[None],
                        n=cat('n0')[None], C=cat('C')[None], bg_chi=d['bg_chi'][None], bg_eta=d['bg_eta'][None],
                        bg_sca=sca[None], phi=phi, wphi=wphi).validate(prob)
    return prob, block, d


def profile_inputs(prob, raw, with_vlos=None):
    """what Engine.set_line_profiles takes for the file's column: (aDamp [1][Nlines][Nspace], vBroad [1][Natoms][Nspace],
    vlos [1][Nspace] or None).  with_vlos: None = hand vlos over iff the context is not phi_compact"""
    lines = [kr for kr, t in enumerate(prob.trans) if t.is_line]
    aD = np.stack([raw['t%d_aDamp' % kr] for kr in lines])[None]
    vB = np.stack([raw['a%d_vBroad' % a] for a in range(prob.Natoms)])[None]
    use = (not prob.phi_compact) if with_vlos is None else with_vlos
    return aD, vB, (np.asarray(raw['vlos'], dtype=np.float64)[None] if use else None)


def gamma_from_raw(d, tag, prob):
    """concatenate the per-atom Gamma snapshots of a golden file into LSX_GAMMA layout"""
    return np.concatenate([d['%s_Gamma_a%d' % (tag, a)].reshape(-1, prob.Nspace) for a in range(prob.Natoms)], axis=0)


def pops_from_raw(d, tag, prob):
    return np.concatenate([d['%s_n_a%d' % (tag, a)] for a in range(prob.Natoms)], axis=0)
