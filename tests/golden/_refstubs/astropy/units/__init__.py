"""Minimal astropy.units stand-in: just enough unit algebra for the reference's
atmosphere constructors (attach with `ndarray << unit`, convert with
`Quantity << unit`, `.value`, `.unit.is_equivalent`)."""
import types
import numpy as np

_NBASE = 5  # m, kg, s, K, sr


class Unit:
    __array_ufunc__ = None  # make ndarray.__lshift__/__mul__ defer to us

    def __init__(self, scale, dims):
        self.scale = float(scale)
        self.dims = tuple(dims)

    def __mul__(self, other):
        if isinstance(other, Unit):
            return Unit(self.scale * other.scale, [a + b for a, b in zip(self.dims, other.dims)])
        return NotImplemented

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return Unit(self.scale / other.scale, [a - b for a, b in zip(self.dims, other.dims)])
        return NotImplemented

    def __pow__(self, p):
        return Unit(self.scale ** p, [a * p for a in self.dims])

    def is_equivalent(self, other):
        if isinstance(other, (list, tuple)):
            return any(self.is_equivalent(o) for o in other)
        return self.dims == other.dims

    def __rlshift__(self, other):
        if isinstance(other, Quantity):
            return other.to(self)
        return Quantity(np.array(other, dtype=np.float64, copy=True), self)

    def __rmul__(self, other):
        return self.__rlshift__(other)

    def __repr__(self):
        return 'Unit(%g, %s)' % (self.scale, self.dims)


class Quantity(np.ndarray):
    def __new__(cls, value, unit):
        obj = np.asarray(value, dtype=np.float64).view(cls)
        obj.unit = unit
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.unit = getattr(obj, 'unit', None)

    @property
    def value(self):
        return np.asarray(self).view(np.ndarray)

    def to(self, unit, equivalencies=None):
        if not self.unit.is_equivalent(unit):
            raise ValueError('incompatible units')
        factor = self.unit.scale / unit.scale
        return Quantity(self.value * factor, unit)

    def __lshift__(self, unit):
        if isinstance(unit, Unit):
            return self.to(unit)
        return NotImplemented


def _base(i, scale=1.0):
    d = [0] * _NBASE
    if i is not None:
        d[i] = 1
    return Unit(scale, d)


m = _base(0)
cm = _base(0, 1e-2)
km = _base(0, 1e3)
nm = _base(0, 1e-9)
kg = _base(1)
g = _base(1, 1e-3)
s = _base(2)
K = _base(3)
sr = _base(4)
one = _base(None)
Hz = s ** -1
J = kg * m ** 2 / s ** 2


def quantity_input(*args, **kwargs):
    def wrap(fn):
        return fn
    return wrap


def spectral_density(wav):
    raise NotImplementedError('stub')


quantity = types.SimpleNamespace(Quantity=Quantity)
