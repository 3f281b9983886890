"""Brute-force NLTE temperature response function (response_fn.py:23-74) as ONE batch of columns:
the base column plus, for every perturbed depth k, the atmosphere with T[k] +/- dT/2, all warm
started from the converged base populations (response_fn.py:33).  The 2 x Nspace MALI solves the
reference runs one after the other are independent columns here.

Set-up quantities of a temperature perturbation at depth k differ from the base column at depth k
only (every set-up formula is pointwise in depth and the uniform height shift of
atmosphere.py:104-105 cancels in |z_k - z_k+1|), so a perturbed column is the base
column plus a "delta": the depth-k entries of the arrays that change."""
import numpy as np

from .problem import ColumnBlock, Problem

# names of the per-column arrays a delta may touch and where they live in a ColumnBlock
_PER_ATOM = {'nStar': 'nStar', 'C': 'C', 'nTotal': 'nTotal'}


def apply_delta(prob: Problem, base: ColumnBlock, delta: dict, k: int, start_n=None) -> ColumnBlock:
    """base: ColumnBlock with ncol == 1.  delta: {array name as in the problem-file format
    (fixtures.py): values at depth k}.  Returns the perturbed column."""
    out = {f: np.array(getattr(base, f), copy=True) for f in ('height', 'temperature', 'nStar', 'nTotal', 'n', 'C',
                                                               'bg_chi', 'bg_eta', 'bg_sca', 'phi', 'wphi')}
    lev_off, lev2_off = prob.lev_off, prob.lev2_off
    line_idx, phi_off = {}, {}
    o = li = 0
    for kr, t in enumerate(prob.trans):
        if t.is_line:
            line_idx[kr], phi_off[kr] = li, o
            o += t.Nlambda
            li += 1
    for key, v in delta.items():
        v = np.asarray(v)
        if key == 'temperature':
            out['temperature'][0, k] = v
        elif key in ('bg_chi', 'bg_eta'):
            out[key][0, :, k] = v
        elif key == 'bg_sca':
            if prob.sca_per_lambda:
                out['bg_sca'][0, :, k] = v
            else:
                out['bg_sca'][0, k] = v
        elif key[0] == 'a' and '_' in key:
            a, name = key[1:].split('_', 1)
            a = int(a)
            nl = prob.Nlevel[a]
            if name == 'nStar':
                out['nStar'][0, lev_off[a]:lev_off[a] + nl, k] = v
            elif name == 'nTotal':
                out['nTotal'][0, a, k] = v
            elif name == 'C':
                out['C'][0, lev2_off[a]:lev2_off[a] + nl * nl, k] = v.reshape(-1)
            # vBroad / weight only matter through phi, which the delta carries explicitly
        elif key[0] == 't' and '_' in key:
            kr, name = key[1:].split('_', 1)
            kr = int(kr)
            if name == 'wphi':
                out['wphi'][0, line_idx[kr], k] = v
            elif name == 'phi':
                t = prob.trans[kr]
                sl = slice(phi_off[kr], phi_off[kr] + t.Nlambda)
                if prob.phi_compact:
                    out['phi'][0, sl, k] = v
                else:
                    out['phi'][0, sl, :, :, k] = v if v.ndim == 3 else v[:, None, None]
    if start_n is not None:
        out['n'][0] = start_n
    return ColumnBlock(**out).validate(prob)
