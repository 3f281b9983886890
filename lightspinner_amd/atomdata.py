"""Atomic data for the library's own set-up chain (lsx_set_atomic_data / lsx_set_atmosphere).

`AtomicData` is the flat description the C ABI takes.  It is built either from Lightspinner-shaped model objects
(`from_models`: duck-typed reads of what atomic_model.AtomicModel / VoigtLine / collisional_rates.* hold) or from the
numeric dump tests/golden/make_golden.py writes (`from_fixture`)."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

from . import _capi
from ._capi import f64

_KINDS = {'Omega': _capi.LSX_COLL_OMEGA, 'CI': _capi.LSX_COLL_CI, 'CE': _capi.LSX_COLL_CE}


@dataclass
class CollisionData:
    kind: int
    i: int
    j: int
    temperature: np.ndarray
    rates: np.ndarray


@dataclass
class LineData:
    i: int
    j: int
    gRad: float
    stark: float
    vdw_kind: int = 0                 # 1: Unsold (atomic_model.py:166-198)
    vdw: Sequence[float] = (0.0, 0.0)


@dataclass
class AtomData:
    weight: float
    is_hydrogen: bool
    E_SI: np.ndarray
    g: np.ndarray
    stage: np.ndarray
    lines: List[LineData] = field(default_factory=list)
    collisions: List[CollisionData] = field(default_factory=list)


@dataclass
class AtomicData:
    atoms: List[AtomData]
    weight_H: float
    weight_He: float
    abundance_He: float

    def to_c(self):
        """-> (LsxAtomicData, keepalive list)"""
        keep = []
        arr = (_capi.LsxAtomModel * len(self.atoms))()
        for a, atom in enumerate(self.atoms):
            nl = len(atom.g)
            lev = (_capi.LsxLevel * nl)()
            for q in range(nl):
                lev[q] = _capi.LsxLevel(float(atom.E_SI[q]), float(atom.g[q]), int(atom.stage[q]), 0)
            lines = (_capi.LsxLineModel * max(1, len(atom.lines)))()
            for q, l in enumerate(atom.lines):
                lines[q] = _capi.LsxLineModel(int(l.i), int(l.j), float(l.gRad), float(l.stark), int(l.vdw_kind), 0,
                                              (C.c_double * 2)(float(l.vdw[0]), float(l.vdw[1])))
            colls = (_capi.LsxCollision * max(1, len(atom.collisions)))()
            for q, k in enumerate(atom.collisions):
                T, R = f64(k.temperature), f64(k.rates)
                keep += [T, R]
                colls[q] = _capi.LsxCollision(int(k.kind), int(k.i), int(k.j), int(T.shape[0]), _capi._ptr(T), _capi._ptr(R))
            arr[a] = _capi.LsxAtomModel(float(atom.weight), 1 if atom.is_hydrogen else 0, nl, lev, len(atom.lines),
                                        len(atom.collisions), lines, colls)
            keep += [lev, lines, colls]
        d = _capi.LsxAtomicData(len(self.atoms), 0, arr, float(self.weight_H), float(self.weight_He), float(self.abundance_He))
        keep.append(arr)
        return d, keep


def from_models(models, line_filter=None) -> AtomicData:
    """models: Lightspinner AtomicModel-shaped objects in active-atom order.  Reads name, atomicTable[...].weight /
    .abundance, levels[].E_SI / g / stage, lines[].i / j / gRad / stark / vdw.vals, collisions[] (class name Omega / CI /
    CE, i, j, temperature, rates).  line_filter(model, line) -> bool selects the lines that are in the transition table."""
    atoms = []
    table = models[0].atomicTable
    for m in models:
        lines = []
        for l in m.lines:
            if line_filter is not None and not line_filter(m, l):
                continue
            vals = list(getattr(getattr(l, 'vdw', None), 'vals', []) or [])
            kind = 1 if type(getattr(l, 'vdw', None)).__name__ == 'VdwUnsold' else 0
            lines.append(LineData(int(l.i), int(l.j), float(l.gRad), float(l.stark), kind, (vals + [0.0, 0.0])[:2]))
        colls = []
        for k in m.collisions:
            kind = _KINDS.get(type(k).__name__)
            if kind is None:
                raise ValueError('collision recipe %s is not supported' % type(k).__name__)
            i, j = int(min(k.i, k.j)), int(max(k.i, k.j))
            colls.append(CollisionData(kind, i, j, np.asarray(k.temperature, dtype=np.float64), np.asarray(k.rates, dtype=np.float64)))
        atoms.append(AtomData(weight=float(table[m.name].weight), is_hydrogen=m.name.upper().strip() == 'H',
                              E_SI=np.array([l.E_SI for l in m.levels]), g=np.array([l.g for l in m.levels]),
                              stage=np.array([l.stage for l in m.levels]), lines=lines, collisions=colls))
    return AtomicData(atoms, float(table['H'].weight), float(table['He'].weight), float(table['He'].abundance))


def from_fixture(d, atoms=None) -> AtomicData:
    """d: tests/golden/setup_falc.npz (make_golden.py setup).  atoms: indices of the fixture's models to take (default all)"""
    names = [str(x) for x in d['atom_names']]
    take = range(len(names)) if atoms is None else atoms
    out = []
    for a in take:
        pre = 'm%d_' % a
        lines = [LineData(int(d[pre + 'line_i'][q]), int(d[pre + 'line_j'][q]), float(d[pre + 'line_gRad'][q]),
                          float(d[pre + 'line_stark'][q]), int(d[pre + 'line_vdw_unsold'][q]), tuple(d[pre + 'line_vdw_vals'][q]))
                 for q in range(d[pre + 'line_i'].shape[0])]
        colls = [CollisionData(int(d[pre + 'col_kind'][q]), int(d[pre + 'col_i'][q]), int(d[pre + 'col_j'][q]),
                               d[pre + 'col_T'][q, :int(d[pre + 'col_nT'][q])], d[pre + 'col_rates'][q, :int(d[pre + 'col_nT'][q])])
                 for q in range(d[pre + 'col_kind'].shape[0])]
        out.append(AtomData(weight=float(d[pre + 'weight']), is_hydrogen=names[a].upper().strip() == 'H', E_SI=d[pre + 'lev_E_SI'],
                            g=d[pre + 'lev_g'], stage=d[pre + 'lev_stage'], lines=lines, collisions=colls))
    return AtomicData(out, float(d['weight_H']), float(d['weight_He']), float(d['abundance_He']))
