"""CHECKER (tests only): line absorption profiles the way the reference defines them -- Voigt function from scipy's
Faddeeva routine (utils.py:13-15), Doppler shift -/+ mu vlos / vBroad for the down / up direction, normalisation
wphi = 1 / sum(phi wlambda wmu / 2) (rh_method.py:198-243) -- evaluated with whole-array numpy operations.  The product
builds its profiles on the device (lsx_set_line_profiles); this file is what the tests compare it with where the
reference's own profiles are not in a fixture."""
import numpy as np
from scipy.special import wofz

C_LIGHT = 2.99792458E+08


def trapezoid_weights(wavelength, lambda0=None):
    """quadrature weight of every point of a transition's local grid (rh_method.py:157-196); lines: in Doppler units"""
    w = np.gradient(np.asarray(wavelength, dtype=np.float64))       # interior: half the distance between the neighbours
    w[0] = 0.5 * (wavelength[1] - wavelength[0])
    w[-1] = 0.5 * (wavelength[-1] - wavelength[-2])
    return w * (C_LIGHT / lambda0 if lambda0 else 1.0)


def profiles(wavelength, lambda0, aDamp, vBroad, vlos, muz, wmu):
    """-> phi [..., Nl, Nrays, 2, Nspace], wphi [..., Nspace] for aDamp, vBroad, vlos of shape [..., Nspace]"""
    lam = np.asarray(wavelength, dtype=np.float64)
    a, vb, vl = np.broadcast_arrays(*(np.asarray(x, dtype=np.float64) for x in (aDamp, vBroad, vlos)))
    muz, wmu = np.asarray(muz, dtype=np.float64), np.asarray(wmu, dtype=np.float64)
    v = ((lam - lambda0) * C_LIGHT)[:, None, None, None] / (vb * lambda0)[..., None, None, None, :]          # [..., Nl, 1, 1, Ns]
    shift = (muz[:, None, None] * np.array([-1.0, 1.0])[None, :, None]) * (vl / vb)[..., None, None, None, :]  # [..., 1, Nrays, 2, Ns]
    phi = wofz(v + shift + 1j * a[..., None, None, None, :]).real / (np.sqrt(np.pi) * vb)[..., None, None, None, :]
    wt = trapezoid_weights(lam, lambda0)[:, None, None, None] * (0.5 * wmu)[None, :, None, None]
    return phi, 1.0 / np.sum(phi * wt, axis=(-4, -3, -2))
