"""Where the float64 tolerances of this repository come from.

SURVEY 8d asks 1e-12 on I and J of one formal solution.  That holds wherever no ray crosses an interval whose optical depth lies
just above the 5e-4 Taylor switch of w2 (formal_solver.py:36-43): there w1 = (1 - e) - dtau e cancels to dtau^2 / 2 and ONE ulp of
exp(-dtau) is up to 9e-10 of that interval's contribution.  The bars stated elsewhere (3e-11 for FALC Ca+H and the 15-level atoms,
1e-11 + envelope for the toy topologies) are not fitted to a measurement any more: tests/envelope.py runs the oracle with every
exp(-dtau) moved one ulp up and one ulp down and every comparison in this file is entry by entry against
    base |x| + 3 |x(+1 ulp) - x(-1 ulp)|,    base = 1e-12 (the rounding-level bar of SURVEY 8d).
CPU: the REFERENCE's own golden vectors lie inside the oracle's envelope -- numpy's SIMD exp against libm's is exactly such a pair --
for all five atoms, and the envelope alone explains the stated bars.  GPU: the HIP kernels on both mappings likewise."""
import numpy as np
import pytest

import envelope
from conftest import golden
from lightspinner_amd import fixtures, synth, Engine, _capi

FIXTURES = ['falc_ca.npz', 'falc_cah.npz', 'falc_c.npz', 'falc_fe.npz', 'falc_mg.npz']
# the single-call bar each fixture is tested with elsewhere (test_oracle_golden.py, test_hip_parity.py, test_production_classes.py)
STATED = {'falc_ca.npz': 1e-12, 'falc_cah.npz': 3e-11, 'falc_c.npz': 3e-11, 'falc_fe.npz': 3e-11, 'falc_mg.npz': 3e-11}


def _runs(oracle_lib, prob, block, ncalls=1, threads=1, profiles=None):
    def make():
        e = Engine(prob, block.ncol, lib=oracle_lib)
        e.set_columns(0, block)
        if profiles is not None:
            e.set_line_profiles(0, *profiles)
        oracle_lib.dll.lsx_oracle_set_threads(e._h, threads)
        return e
    return envelope.oracle_runs(oracle_lib, make, ncalls, what=(_capi.LSX_I, _capi.LSX_J))


@pytest.mark.parametrize('name', FIXTURES)
def test_reference_lies_inside_the_one_ulp_exp_envelope_of_the_oracle(oracle_lib, name):
    prob, block, d = fixtures.load_problem_npz(golden(name))
    runs = _runs(oracle_lib, prob, block)
    rel_I, env_I = envelope.inside(d['fs1_I'][None], runs, 0, _capi.LSX_I, 1e-12)
    rel_J, env_J = envelope.inside(d['fs1_J'][None], runs, 0, _capi.LSX_J, 1e-12)
    # the hook does something (some ray of every FALC problem has an interval in the exponential's regime) ...
    assert env_I > 1e-13
    # ... and the stated bar of this fixture is what the envelope gives, not more than a factor of ~ 10 above it and not below the
    # reference's own deviation: CaII 2.3e-12 (bar 1e-12 + ...: measured deviation 2.4e-13), Ca+H 2.4e-11 (bar 3e-11, deviation 8e-12)
    assert max(rel_I, rel_J) < STATED[name]
    assert STATED[name] < 20 * max(env_I, 1e-12), (env_I, STATED[name])


def test_the_hook_moves_w2_by_one_ulp_of_the_exponential_only(oracle_lib):
    w = np.zeros(2)
    out = {}
    for ulp in (0, 1, -1):
        oracle_lib.dll.lsx_oracle_set_exp_ulp(ulp)
        vals = []
        for x in (1e-4, 6e-4, 0.3, 20.0, 60.0):
            oracle_lib.dll.lsx_oracle_w2(float(x), _capi._ptr(w))
            vals.append(w.copy())
        out[ulp] = np.array(vals)
    oracle_lib.dll.lsx_oracle_set_exp_ulp(0)
    # Taylor branch and saturated branch: untouched; the exponential's branch: w0 = 1 - e moves by an ulp of e, w1 likewise
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][4], out[0][4])
    for i, x in ((1, 6e-4), (2, 0.3), (3, 20.0)):
        ulp_e = np.spacing(np.exp(-x))
        assert abs((out[1][i][0] - out[-1][i][0]) + 2 * ulp_e) <= 2.3e-16      # (1 - e rounds to the grid of 1: half an ulp of 1 each side)
        assert out[1][i][1] <= out[0][i][1] <= out[-1][i][1]
    # just above the switch the relative effect on w1 is the 1e-9 of DESIGN.md 2
    assert 1e-10 < (out[-1][1][1] - out[1][1][1]) / out[0][1][1] < 3e-9


@pytest.mark.gpu
@pytest.mark.parametrize('policy', ['ray-per-lane', 'ray-serial'])
@pytest.mark.parametrize('name,ncol', [('falc_ca.npz', 35), ('falc_cah.npz', 33)])
def test_hip_lies_inside_the_envelope_on_production_columns(hip_lib, oracle_lib, name, ncol, policy):
    """FALC-perturbed columns with a line-of-sight velocity (C3 / C4 shape), per-class launches on both mappings: every I and J of the
    first formal solution inside 1e-12 + 3 x the envelope -- ray by ray, not in the maximum norm"""
    prob, base, raw = fixtures.load_problem_npz(golden(name), phi_compact=False)
    blk, prof = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=97, vlos_sigma=2.0e3)
    runs = _runs(oracle_lib, prob, blk, threads=8, profiles=prof)
    e = Engine(prob, ncol, lib=hip_lib, sweep_policy=policy)
    synth.load_columns(e, blk, prof)
    e.formal_sol_gamma()
    rel_I, env_I = envelope.inside(e.get(_capi.LSX_I), runs, 0, _capi.LSX_I, 1e-12)
    rel_J, env_J = envelope.inside(e.get(_capi.LSX_J), runs, 0, _capi.LSX_J, 1e-12)
    assert max(rel_I, rel_J) < STATED[name]
    e.close()
