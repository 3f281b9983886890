"""Wavelength grid and active set: the counterpart of RadiativeSet.compute_wavelength_grid
(atomic_set.py:377-455) on Lightspinner-shaped model objects, evaluated by the library (lsx_wavelength_grid,
lsx_active_set, lsx_continuum_alpha -- host C++ inside liblsx_hip.so).

    spect = compute_wavelength_grid(models)            # sets t.wavelength (and t.alpha on continua), like the reference
    ctx = Context(atmos, spect, eqPops, background)

Reads from every model: .lines / .continua; from a line .wavelength; from a continuum .wavelength, .lambdaEdge and either
(.alpha0, .minLambda, .iLevel.E_SI, .jLevel.E_SI, .jLevel.stage) -- hydrogenic -- or the tabulated (.wavelength, .alpha)."""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _capi


def _is_line(t):
    return hasattr(t, 'Aji') and hasattr(t, 'lambda0') and not hasattr(t, 'lambdaEdge')


@dataclass
class ActiveAtoms:
    """the one attribute of the reference's RadiativeSet that Context reads (rh_method.py:557)"""
    activeAtoms: List[object]


@dataclass
class SpectrumConfiguration:
    """atomic_set.py:40-53, the fields Context and ComputationalTransition read"""
    radSet: ActiveAtoms
    wavelength: np.ndarray
    transitions: List[object]
    models: List[object]
    blueIdx: List[int]
    activeSet: List[List[object]]
    redIdx: List[int] = field(default_factory=list)
    active: Optional[np.ndarray] = None            # bool [Ntrans][Nspect], rows in `transitions` order


def continuum_alpha(cont, wavelength, lib=None):
    """cont.compute_alpha(wavelength) (atomic_model.py:606-612, 662-671) through the library"""
    lib = lib or _capi.load_hip_library()
    if hasattr(cont, 'alpha0'):
        return lib.continuum_alpha(wavelength, edge=cont.lambdaEdge, min_lambda=cont.minLambda, alpha0=cont.alpha0,
                                   E_i=cont.iLevel.E_SI, E_j=cont.jLevel.E_SI, stage_j=cont.jLevel.stage)
    table = (np.asarray(cont.wavelength, dtype=np.float64), np.asarray(cont.alpha, dtype=np.float64))
    return lib.continuum_alpha(wavelength, edge=cont.lambdaEdge, min_lambda=table[0][0], table=table)


def compute_wavelength_grid(models, extraWavelengths=None, lambdaReference=500.0, lib=None) -> SpectrumConfiguration:
    """models: the active (and detailed-LTE) atoms, in the order the transition table shall have.  As in the reference,
    every transition's .wavelength is replaced by its slice of the merged grid and every continuum's .alpha by the
    cross-section on that slice."""
    lib = lib or _capi.load_hip_library()
    models = list(models)
    if not models:
        raise ValueError('Need at least one atom active or in detailed LTE')           # atomic_set.py:378-379
    transitions = [t for m in models for t in (list(m.lines) + list(m.continua))]        # :392-399
    is_line = [_is_line(t) for t in transitions]
    grids = [np.asarray(t.wavelength, dtype=np.float64) for t in transitions]
    edges = [0.0 if l else float(t.lambdaEdge) for t, l in zip(transitions, is_line)]
    wavelength, blue, red = lib.wavelength_grid(grids, is_line, edges, extra=extraWavelengths, lambda_reference=lambdaReference)
    active = lib.active_set(blue, red, wavelength.shape[0])
    for t, l, b, r in zip(transitions, is_line, blue, red):                              # :409-416
        w = wavelength[b:r].copy()
        if not l:
            t.alpha = continuum_alpha(t, w, lib=lib)
        t.wavelength = w
    activeSet = [[t for kr, t in enumerate(transitions) if active[kr, la]] for la in range(wavelength.shape[0])]
    return SpectrumConfiguration(radSet=ActiveAtoms(models), wavelength=wavelength, transitions=transitions, models=models,
                                 blueIdx=[int(b) for b in blue], activeSet=activeSet, redIdx=[int(r) for r in red], active=active)
