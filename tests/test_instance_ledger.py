"""Ledger of the compiled sweep instances (CPU side; the GPU side is tests/test_instances_gpu.py, the shared data
tests/instance_cases.py): an instance that lsx_plan.h lists is compiled into the product and can be dispatched to, so some GPU parity
test has to plan it.  The plans come from the product's own planner (lsx_plan.cpp built into liblsx_host.so, `make host`)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden
import instance_cases as ic
from lightspinner_amd import fixtures

CSRC = os.path.join(ROOT, 'lightspinner_amd', 'csrc')
pytestmark = pytest.mark.skipif(shutil.which('g++') is None, reason='no host compiler')


@pytest.fixture(scope='module')
def host():
    subprocess.check_call(['make', '-s', '-C', CSRC, 'host'])
    from san_driver import HostOnly
    h = HostOnly(os.path.join(CSRC, 'liblsx_host.so'))
    h.dll.lsx_plan_instances.restype = C.c_int32
    h.dll.lsx_plan_instances.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    h.dll.lsx_plan_rs_classes.restype = C.c_int32
    return h


def _instances(host, which):
    buf = (C.c_int32 * 64)()
    n = host.dll.lsx_plan_instances(which, buf, 64)
    assert 0 < n <= 64
    return [ic.decode(int(buf[i])) for i in range(n)]


def _planned(host, prob, bits=0):
    rc, msg, s, tiles = host.probe(prob, bits)
    assert rc == 0, msg
    return {ic.decode(int(t[6])) for t in tiles}


def _rs_classes(host, prob, bit=1):
    p, keep = prob.to_c()
    out = (C.c_int32 * 128)()
    n = host.dll.lsx_plan_rs_classes(C.byref(p), out, 64)
    assert n > 0
    return {ic.decode(int(out[2 * i])) for i in range(n) if out[2 * i + 1] & bit}


def test_every_compiled_instance_is_planned_by_a_gpu_parity_test(host):
    lane, serial = _instances(host, 0), _instances(host, 1)
    assert set(serial) <= set(lane) and all(k[0] <= 2 for k in serial)
    # ---- what the GPU suite's problems plan
    by_case = {}
    for name, ncol, Ns, compact in ic.CASES:
        prob, _ = ic.build(name, 2, Ns, compact)
        got = _planned(host, prob)
        assert set(ic.EXPECT[name]) <= got, (name, sorted(got))
        by_case[name] = (got, _rs_classes(host, prob), _rs_classes(host, prob, 2))
    # the reference's own problems (tests/test_production_classes.py: both kernels; the parabolic rule: tests/parabolic_cases.py)
    # ... and the reference's larger atoms (round 5: carbon, iron, MgII -- tests/test_production_classes.py,
    # test_multilevel_reference_atoms_per_class_path): linked to one, two and three lines; with the linking off their continua become
    # per-ray slots -- per-ray-continuum instances planned by a REFERENCE problem
    for fx in ('falc_ca.npz', 'falc_cah.npz', 'falc_c.npz', 'falc_fe.npz', 'falc_mg.npz', 'falc_all.npz'):
        prob, base, raw = fixtures.load_problem_npz(golden(fx), phi_compact=False)
        by_case[fx] = (_planned(host, prob), _rs_classes(host, prob), _rs_classes(host, prob, 2))
        if fx in ('falc_c.npz', 'falc_fe.npz', 'falc_mg.npz', 'falc_all.npz'):
            by_case[fx + ':unlinked'] = (_planned(host, prob, 1), set(), set())
    assert (2, 1, 0, 0) in by_case['falc_c.npz:unlinked'][0] and (4, 1, 0, 0) in by_case['falc_mg.npz:unlinked'][0]
    assert (3, 3, 1, 0) in by_case['falc_mg.npz'][0]
    assert (2, 2, 0, 2) in by_case['falc_all.npz'][0] and (3, 1, 0, 0) in by_case['falc_all.npz:unlinked'][0]      # all five atoms active
    lane_hit = set().union(*(g for g, _, _ in by_case.values()))
    serial_hit = set().union(*(r for _, r, _ in by_case.values()))
    # the ray-serial instances of the parabolic rule (tests/test_instances_gpu.py, mode 'parabolic-serial'): a subset of the ray-serial list
    par_serial = _instances(host, 2)
    assert set(par_serial) <= set(serial) and set(par_serial) == set(ic.PARABOLIC_SERIAL)
    par_hit = set().union(*(r for _, _, r in by_case.values()))
    assert [k for k in par_serial if k not in par_hit] == [] and par_hit <= set(par_serial)
    assert [k for k in lane if k not in lane_hit] == [], 'one-ray-per-lane instances no GPU test plans'
    # the ray-serial kernel and the parabolic rule's compile-time classes share their list (lsx_plan.h)
    assert [k for k in serial if k not in serial_hit] == [], 'ray-serial / parabolic instances no GPU test plans'
    # every class a case reaches on the ray-serial kernel has an instance there (nothing falls through to a wrong one)
    assert serial_hit <= set(serial)
    # the generic instances (runtime slot loops), with and without linked continua
    assert {(-1, 0, 0, 0), (-1, 0, 1, 0)} <= lane_hit


def test_shapes_without_an_instance_cannot_be_planned(host):
    """the shapes dropped from the lists in round 4 -- (1,0) (2,0): per-ray slots without a line; (2,1,linked) (3,1,linked): linked
    continua beside a single line whose atom has a per-ray continuum -- do not occur in any plan: 400 random transition tables,
    the reference's problems, the toy topologies, with and without linking"""
    from san_driver import random_problem
    from toy import toy_problem
    rng = np.random.default_rng(2024)
    lane = set(_instances(host, 0))
    seen = set()
    probs = [random_problem(rng) for _ in range(400)]
    probs += [toy_problem(seed=s, Nrays=5, Nspect=120, ncol=1, **kw)[0] for s, kw in ((1, {}), (2, dict(chain=False)), (3, dict(multiplet=3)), (4, dict(multiplet=4)))]
    probs += [fixtures.load_problem_npz(golden(fx), phi_compact=False)[0] for fx in ('falc_ca.npz', 'falc_cah.npz')]
    for prob in probs:
        for bits in (0, 1, 4):          # default, LSX_NO_LINKED, LSX_NO_TOPO
            rc, msg, s, tiles = host.probe(prob, bits)
            if rc != 0:
                assert rc in (1, 5), msg
                continue
            for t in tiles:
                nP, nL, nK = int(t[2]), int(t[4]), int(t[5])
                assert not (nP > 0 and nL == 0), 'per-ray slots without a line'
                assert not (nK > 0 and nL == 1 and nP > 1), 'linked continua beside one line and its atom\'s per-ray continua'
                k = ic.decode(int(t[6]))
                seen.add(k)
                assert k[0] < 0 or k in lane, k
    assert len(seen) >= 12


def test_which_epilogue_the_reference_problems_tiles_take(host):
    """DevTile.fast_simple as the planner files the reference's problems (lsx_plan.cpp): 2 = the column-mapped fast-continuum epilogue's
    plain instances (at most six continua per atom, twelve per tile), 3 = its big-set instances (round 5: carbon's and iron's fourteen,
    MgII's ten bound-free continua onto one level), 1 = the row-mapped kernel.  BASELINE's CaII and Ca+H problems take the plain
    instances only -- their path is untouched by the big-set instances; with all five atoms active 183 of 328 tiles take those."""
    from collections import Counter
    want = {'falc_ca.npz': {2}, 'falc_cah.npz': {2}, 'falc_c.npz': {2, 3}, 'falc_fe.npz': {2, 3}, 'falc_mg.npz': {1, 2, 3}, 'falc_all.npz': {1, 2, 3}}
    for fx, kinds in want.items():
        prob, base, raw = fixtures.load_problem_npz(golden(fx), phi_compact=False)
        rc, msg, s, tiles = host.probe(prob, 0)
        assert rc == 0, msg
        c = Counter(int(t[7]) for t in tiles if t[3] > 0)        # tiles with fast continua
        assert set(c) == kinds, (fx, c)
        if fx == 'falc_all.npz':
            assert c[3] == 183 and c[2] == 136 and c[1] == 9, c
