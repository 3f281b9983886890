"""Drop-in counterpart of Lightspinner's formal_solver.piecewise_linear_1d
(formal_solver.py:144-212), evaluated by the HIP library.

    iPsi = piecewise_linear_1d(atmos, mu, toFrom, wav, chi, S)   # -> IPsi(I, PsiStar)

The reference calls this once per (wavelength, ray, direction) from its Python loop nest; the
engine fuses it into the sweep kernel, so this entry point exists for callers that use the
formal solver on its own.  `piecewise_linear_batch` solves many rays in one launch."""
from dataclasses import dataclass

import numpy as np

from . import _capi


@dataclass
class IPsi:
    """formal_solver.py:6-12"""
    I: np.ndarray
    PsiStar: np.ndarray


def piecewise_linear_batch(atmos, mu_values, to_obs, wav, chi, S, device=0, lib=None):
    """chi, S: [nray][Nspace]; mu_values (cosines), to_obs (bool), wav (nm): [nray]"""
    lib = lib if lib is not None else _capi.load_hip_library()
    return lib.piecewise_linear_1d(atmos.height, atmos.temperature, mu_values, np.asarray(to_obs, dtype=np.int32), wav,
                                   chi, S, device=device)


def piecewise_linear_1d(atmos, mu, toFrom, wav, chi, S, device=0, lib=None) -> IPsi:
    """Same signature and semantics as the reference: `mu` indexes atmos.muz, `toFrom` True = ray
    towards the observer (thermalised lower boundary), False = away (zero incident radiation)."""
    I, Psi = piecewise_linear_batch(atmos, [atmos.muz[mu]], [1 if toFrom else 0], [wav],
                                    np.asarray(chi, dtype=np.float64)[None], np.asarray(S, dtype=np.float64)[None],
                                    device=device, lib=lib)
    return IPsi(I[0], Psi[0])
