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


# ---- models that carry their atomic data (what Lightspinner's AtomicModel / VoigtLine / collisional_rates objects hold),
# rebuilt from tests/golden/setup_falc.npz: Context then takes the library's own set-up chain ------------------------
class _Level:
    def __init__(self, E_SI, g, stage):
        self.E_SI, self.g, self.stage = float(E_SI), float(g), int(stage)


class VdwUnsold:
    def __init__(self, vals):
        self.vals = [float(v) for v in vals]


class VdwNone:
    vals = []


class _TableCollision:
    def __init__(self, i, j, T, R):
        self.i, self.j, self.temperature, self.rates = int(i), int(j), np.array(T), np.array(R)


class Omega(_TableCollision):
    pass


class CI(_TableCollision):
    pass


class CE(_TableCollision):
    pass


class _Element:
    def __init__(self, weight, abundance):
        self.weight, self.abundance = float(weight), float(abundance)


class DataLine(FakeLine):
    """a line that knows gRad / stark / vdw instead of offering damping()"""
    damping = None

    def __init__(self, d, kr, s, m, q):
        FakeLine.__init__(self, d, kr)
        del self._aDamp
        self.gRad, self.stark = float(s['m%d_line_gRad' % m][q]), float(s['m%d_line_stark' % m][q])
        self.vdw = VdwUnsold(s['m%d_line_vdw_vals' % m][q]) if s['m%d_line_vdw_unsold' % m][q] else VdwNone()


class DataAtom:
    v_broad = None

    def __init__(self, d, a, s, table):
        self.name = str(d['atom_names'][a])
        m = [str(x) for x in s['atom_names']].index(self.name)
        pre = 'm%d_' % m
        self.atomicTable = table
        self.levels = [_Level(E, g, st) for E, g, st in zip(s[pre + 'lev_E_SI'], s[pre + 'lev_g'], s[pre + 'lev_stage'])]
        self.kr = [kr for kr in range(len(d['t_atom'])) if d['t_atom'][kr] == a]
        pos = {(int(i), int(j)): q for q, (i, j) in enumerate(zip(s[pre + 'line_i'], s[pre + 'line_j']))}
        self.lines = [DataLine(d, kr, s, m, pos[(int(d['t_i'][kr]), int(d['t_j'][kr]))]) for kr in self.kr if d['t_isline'][kr]]
        self.continua = [FakeContinuum(d, kr) for kr in self.kr if not d['t_isline'][kr]]
        cls = {0: Omega, 1: CI, 2: CE}
        self.collisions = [cls[int(s[pre + 'col_kind'][q])](s[pre + 'col_i'][q], s[pre + 'col_j'][q],
                                                            s[pre + 'col_T'][q, :int(s[pre + 'col_nT'][q])],
                                                            s[pre + 'col_rates'][q, :int(s[pre + 'col_nT'][q])])
                           for q in range(s[pre + 'col_kind'].shape[0])]


def build_data_fakes(d, s, start_pops=None):
    """like build_fakes, with models that hold atomic data (d: a problem fixture, s: setup_falc.npz)"""
    names = [str(x) for x in s['atom_names']]
    table = {n: _Element(s['m%d_weight' % m], s['m%d_abundance' % m]) for m, n in enumerate(names)}
    table['H'] = _Element(s['weight_H'], 1.0)
    table['He'] = _Element(s['weight_He'], s['abundance_He'])
    atmos = FakeAtmos(d)
    atoms = [DataAtom(d, a, s, table) for a in range(len(d['atom_names']))]
    spect = FakeSpect(d, atoms)
    eq = FakePops()
    for a, atom in enumerate(atoms):
        eq[atom.name] = FakeState(np.array(d['a%d_nStar' % a]), np.array(d['a%d_nTotal' % a]),
                                  None if start_pops is None else np.array(start_pops[a]))
    if 'H' not in eq:
        eq['H'] = FakeState(np.array(d['hGround'])[None], None)
    return atmos, spect, eq, FakeBackground(d)
