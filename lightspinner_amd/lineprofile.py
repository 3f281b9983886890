"""Host-side setup of line absorption profiles (inputs of the hot path).

Restates ComputationalTransition.compute_phi / wlambda (rh_method.py:157-243) in
batched numpy so that many columns can be prepared at once; the Voigt function is
the same scipy Faddeeva the reference uses (utils.py:13-15).  Setup only: runs once
per Context, never inside the MALI iteration.
"""
import numpy as np
from scipy.special import wofz

from . import constants as Const


def wlambda(wavelength, lambda0=None):
    """rh_method.py:157-196 (array form): trapezoid weights on the local grid,
    times c/lambda0 for lines (Doppler units), times 1 for continua."""
    wavelength = np.asarray(wavelength, dtype=np.float64)
    dopplerWidth = Const.CLight / lambda0 if lambda0 else 1.0
    wla = np.zeros_like(wavelength)
    wla[0] = 0.5 * (wavelength[1] - wavelength[0])
    wla[-1] = 0.5 * (wavelength[-1] - wavelength[-2])
    wla[1:-1] = 0.5 * (wavelength[2:] - wavelength[:-2])
    return dopplerWidth * wla


def voigt_H(a, v):
    """utils.py:13-15"""
    return wofz(v + 1j * a).real


def compute_phi(wavelength, lambda0, aDamp, vBroad, vlos, muz, wmu):
    """rh_method.py:198-243.

    aDamp, vBroad, vlos: [..., Nspace] (any leading batch of columns)
    returns phi [..., Nl, Nrays, 2, Nspace], wphi [..., Nspace]
    """
    wavelength = np.asarray(wavelength, dtype=np.float64)
    aDamp = np.asarray(aDamp, dtype=np.float64)
    vBroad = np.asarray(vBroad, dtype=np.float64)
    vlos = np.asarray(vlos, dtype=np.float64)
    muz = np.asarray(muz, dtype=np.float64)
    wmu = np.asarray(wmu, dtype=np.float64)
    sqrtPi = np.sqrt(np.pi)
    Nl, Nrays = wavelength.shape[0], muz.shape[0]
    batch = np.broadcast(aDamp, vBroad, vlos).shape
    phi = np.zeros(batch[:-1] + (Nl, Nrays, 2, batch[-1]))
    wPhi = np.zeros(batch)
    wLambda = wlambda(wavelength, lambda0)
    vlosDop = [muz[mu] * vlos / vBroad for mu in range(Nrays)]
    for la in range(Nl):
        v = (wavelength[la] - lambda0) * Const.CLight / (vBroad * lambda0)
        for mu in range(Nrays):
            wlamu = wLambda * 0.5 * wmu[mu]
            for toFrom, sign in enumerate([-1.0, 1.0]):
                vk = v + sign * vlosDop[mu]
                p = voigt_H(aDamp, vk) / (sqrtPi * vBroad)
                phi[..., la, mu, toFrom, :] = p
                wPhi += p * wlamu[la]
    return phi, 1.0 / wPhi
