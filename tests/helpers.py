"""Duck-typed stand-ins for Lightspinner's Atmosphere / SpectrumConfiguration / AtomicStateTable /
Background / AtomicModel objects, rebuilt from a golden fixture (the reference itself is absent on
the GPU box).  They expose exactly the attributes rh_method.Context reads."""
import numpy as np


class FakeAtmos:
    def __init__(self, d):
        for k in ('height', 'temperature', 'ne', 'vlos', 'vturb', 'nHTot', 'muz', 'wmu'):
            setattr(self, k, np.array(d[k], dtype=np.float64))
        self.Nspace = self.height.shape[0]
        self.Nrays = self.muz.shape[0]
        self.nondim_calls = 0

    def nondimensionalise(self):
        self.nondim_calls += 1


class FakeLine:
    def __init__(self, d, kr):
        self.i, self.j = int(d['t_i'][kr]), int(d['t_j'][kr])
        self.Aji, self.Bji, self.Bij = float(d['t_Aji'][kr]), float(d['t_Bji'][kr]), float(d['t_Bij'][kr])
        self.lambda0 = float(d['t_lambda0'][kr])
        self.wavelength = np.array(d['t%d_wavelength' % kr])
        self.Nlambda = self.wavelength.shape[0]
        self._aDamp = np.array(d['t%d_aDamp' % kr])

    def damping(self, atmos, vBroad, hGround):
        return self._aDamp, None


class FakeContinuum:
    def __init__(self, d, kr):
        self.i, self.j = int(d['t_i'][kr]), int(d['t_j'][kr])
        self.wavelength = np.array(d['t%d_wavelength' % kr])
        self.alpha = np.array(d['t%d_alpha' % kr])


class FakeCollision:
    def __init__(self, C):
        self._C = C

    def compute_rates(self, atmos, nStar, Cmat):
        Cmat += self._C


class FakeAtom:
    def __init__(self, d, a):
        self.name = str(d['atom_names'][a])
        nl = d['a%d_nStar' % a].shape[0]
        self.levels = list(range(nl))
        self.lines = [FakeLine(d, kr) for kr in range(len(d['t_atom'])) if d['t_atom'][kr] == a and d['t_isline'][kr]]
        self.continua = [FakeContinuum(d, kr) for kr in range(len(d['t_atom'])) if d['t_atom'][kr] == a and not d['t_isline'][kr]]
        self.collisions = [FakeCollision(np.array(d['a%d_C' % a]))]
        self._vBroad = np.array(d['a%d_vBroad' % a])
        self.kr = [kr for kr in range(len(d['t_atom'])) if d['t_atom'][kr] == a]

    def v_broad(self, atmos):
        return self._vBroad


class FakeState:
    def __init__(self, nStar, nTotal, pops=None):
        self.nStar, self.nTotal, self.pops = nStar, nTotal, pops

    @property
    def n(self):
        return self.pops if self.pops is not None else self.nStar


class FakePops(dict):
    atomicTable = None


class FakeRadSet:
    def __init__(self, atoms):
        self.activeAtoms = atoms


class FakeSpect:
    def __init__(self, d, atoms):
        self.wavelength = np.array(d['wavelength'])
        self.radSet = FakeRadSet(atoms)
        ordered = [t for a in atoms for t in (a.lines + a.continua)]
        self.transitions = ordered
        act = d['t_active']
        krs = [kr for a in atoms for kr in a.kr]   # table order == [lines, continua] per atom
        self.activeSet = [[t for t, kr in zip(ordered, krs) if act[kr, la]] for la in range(self.wavelength.shape[0])]


class FakeBackground:
    def __init__(self, d):
        self.chi = np.array(d['bg_chi'])
        self.eta = np.array(d['bg_eta'])
        sca = np.array(d['bg_sca'])
        self.sca = sca if sca.ndim == 2 else np.tile(sca, (self.chi.shape[0], 1))   # background.py:45-47


def build_fakes(d, start_pops=None):
    atmos = FakeAtmos(d)
    atoms = [FakeAtom(d, a) for a in range(len(d['atom_names']))]
    spect = FakeSpect(d, atoms)
    eq = FakePops()
    for a, atom in enumerate(atoms):
        eq[atom.name] = FakeState(np.array(d['a%d_nStar' % a]), np.array(d['a%d_nTotal' % a]),
                                  None if start_pops is None else np.array(start_pops[a]))
    if 'H' not in eq:
        eq['H'] = FakeState(np.array(d['hGround'])[None], None)
    return atmos, spect, eq, FakeBackground(d)
