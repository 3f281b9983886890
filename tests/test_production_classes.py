"""GPU parity at the PRODUCTION launch path and workloads (BASELINE configs C3 / C4): >= 32 columns take one kernel
instantiation per tile class on forked streams (lsx_hip.hip: enqueue_fs), not the fused small-batch kernel the
single-column tests reach.  C4 (Ca+H): hydrogen's bound-free continua overlap its lines; they are "linked" continua
(handled outside the sweep with the line's sum_mu w Psi* phi, lsx_hip.hip: roles_of), so the classes that run are the
line-only ones with and without linked continua -- all of them meet the oracle here; the classes with three and four
per-ray slots (round 1's `lsx_sweep_kernel<3|4, ...>`) run with the linking switched off (options='linked=0').

Inputs: synth.perturbed_columns(..., vlos_sigma=2e3) -- FALC-perturbed columns with a smooth line-of-sight
velocity (2 km/s), so the line profiles are ray dependent and are built by each library's own lsx_set_line_profiles
(the two Voigt functions agree to 3e-14, tests/test_line_profiles.py).

Tolerances: first formal-solution call J, I <= tol relative, off-diagonal Gamma <= 10 tol, diagonal <= tol of
its column's largest entry; tol = 1e-12 (CaII), 3e-11 (Ca+H: DESIGN 2, the w2 cancellation next to the Taylor switch);
after 8 MALI iterations (5 of them with stat_equil) the populations, J and I against bars computed from the oracle on the same columns
(tests/envelope.py, SequenceBars; round 5: a flat 1e-8 / 1e-7)."""
import ctypes as C

import numpy as np
import pytest

import envelope

from conftest import golden, relerr, gamma_err
from lightspinner_amd import fixtures, synth, Engine, _capi

pytestmark = pytest.mark.gpu


def class_table(lib, eng):
    """-> ({(per-ray slots, lines, linked, two-line relation): (tiles, launches)}, fused launches)"""
    f = lib.dll.lsx_hip_class_info
    f.restype = C.c_int32
    f.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    out = (C.c_int64 * 8)()
    n = f(eng._h, -1, out)
    fused = int(out[0])
    table = {}
    class_table.ray_serial = {}
    for i in range(n):
        f(eng._h, i, out)
        table[(int(out[0]), int(out[1]), int(out[4]), int(out[5]))] = (int(out[2]), int(out[3]))
        class_table.ray_serial[(int(out[0]), int(out[1]), int(out[4]), int(out[5]))] = bool(out[6])
    return table, fused


_BARS = {}      # (fixture, columns, seed, iterations) -> envelope.SequenceBars: the two mappings of a case share the oracle's three runs


def _run_pair(hip_lib, oracle_lib, name, ncol, seed, tol, expect_classes, iters=8, **engine_kw):
    """engine_kw: what the HIP engine is made with (sweep_policy=, options=: include/lsx.h, lsx_set_sweep_policy /
    lsx_create_with_options) -- explicit arguments, not the process environment.
    Bars: `tol` on the first call (identical inputs); behind the statistical equilibria they are computed from the oracle on the
    same columns (tests/envelope.py, SequenceBars: +-1-ulp-exp spread through the same calls, the LU's componentwise conditioning,
    the population deviation measured going into a formal solution) -- round 5 asserted a flat 1e-8 / 1e-7 here."""
    prob, base, raw = fixtures.load_problem_npz(golden(name), phi_compact=False)
    blk, (aD, vB, vlos) = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=seed, vlos_sigma=2.0e3)
    assert vlos is not None and np.any(vlos[1] != 0.0)

    # the oracle's side: its plain run IS the reference the HIP engine is compared with, its +-1-ulp-exp runs and the LU's conditioning
    # give the bars -- one set of three runs per ensemble; a case on fewer columns of the same ensemble (the `unlinked` legs: 33 of 36,
    # 40 of 41: perturbed_columns seeds every column by its absolute index) takes the first columns of it
    nbar = {33: 36, 40: 41}.get(ncol, ncol)
    key = (name, nbar, seed, iters)
    if key not in _BARS:
        bblk, (baD, bvB, bvlos) = (blk, (aD, vB, vlos)) if nbar == ncol else synth.perturbed_columns(prob, base, raw, ncol=nbar, seed=seed, vlos_sigma=2.0e3)
        assert np.array_equal(bvlos[:ncol], vlos) and np.array_equal(bblk.n[:ncol], blk.n)

        def make_oracle():
            e = Engine(prob, nbar, lib=oracle_lib)
            e.set_columns(0, bblk)
            e.set_line_profiles(0, baD, bvB, bvlos)
            oracle_lib.dll.lsx_oracle_set_threads(e._h, 16)
            return e
        _BARS[key] = envelope.SequenceBars(oracle_lib, make_oracle, prob, iters, 3, tol)
    full = _BARS[key]
    bars = full if nbar == ncol else full.subset(ncol)
    ora = lambda call, what: bars.oracle(call, what)           # the oracle's plain run, call by call
    hip = Engine(prob, ncol, lib=hip_lib, **engine_kw)
    hip.set_columns(0, blk)
    hip.set_line_profiles(0, aD, vB, vlos)
    engs = [hip]
    # ---- first call: identical inputs on both sides
    dJ = hip.formal_sol_gamma()
    assert dJ == 1.0 and full.oracle(0, 'dJ') == 1.0
    assert relerr(hip.get(_capi.LSX_J), ora(0, _capi.LSX_J)) < tol
    assert relerr(hip.get(_capi.LSX_I), ora(0, _capi.LSX_I)) < tol
    off, diag = gamma_err(hip.get(_capi.LSX_GAMMA), ora(0, _capi.LSX_GAMMA), prob)
    assert off < 10 * tol and diag < tol, (off, diag)
    # per-column monitors agree column by column
    assert np.allclose(hip.get(_capi.LSX_DJ_COL), ora(0, _capi.LSX_DJ_COL), rtol=1e-9)
    # ---- `iters` (8) MALI iterations (test.py:20-29: the first three update J only)
    dn = dn_in = 0.0
    for it in range(2, iters + 1):
        dJ = hip.formal_sol_gamma()
        if nbar == ncol:
            assert dJ == pytest.approx(full.oracle(it - 1, 'dJ'), rel=1e-6)
        if it > 3:
            dP = hip.stat_equil()
            if nbar == ncol:
                assert dP == pytest.approx(full.oracle(it - 1, 'dP'), rel=1e-6)
            dn_in = dn                                          # what the populations differed by going into this iteration's formal solution
            dn = bars.check_n(hip.get(_capi.LSX_N), ora(it - 1, _capi.LSX_N), it - 1, ' (HIP vs oracle, iteration %d)' % it, dn_in)
    bI, eI = bars.I_bar(iters - 1, dn_in), relerr(hip.get(_capi.LSX_I), ora(iters - 1, _capi.LSX_I))
    rJ, eJ, bJ = bars.J_excess(hip.get(_capi.LSX_J), ora(iters - 1, _capi.LSX_J), iters - 1, dn_in)
    print('%s %s: after %d iterations n %.2e (bars per atom %s), J %.2e (%.2f x its bar entry by entry; bar where nothing is transmitted %.1e), I %.2e (bar %.1e)'
          % (name, engine_kw, iters, dn, ['%.1e' % b for b in bars.n_bar(iters - 1, dn_in)], eJ, rJ, bJ, eI, bI))
    assert rJ <= 1.0 and eI < bI, (rJ, eJ, bJ, eI, bI)
    # ---- the production instantiations are what ran
    table, fused = class_table(hip_lib, hip)
    assert fused == 0, 'the fused small-batch kernel must not be what this test measures'
    for key in expect_classes:
        assert key in table and table[key][1] == iters, (key, table)
    assert all(launches == iters for _, launches in table.values())
    for e in engs:
        e.close()
    return table


# Both sweep kernels are production paths: one ray per lane (lsx_sweep.hip; what contexts with few columns run) and the
# ray-serial kernel (lsx_sweep_rs.hip: five columns per wavefront; contexts of 160 columns or more).
# The mapping is pinned with Engine(sweep_policy=...) so that both meet the oracle on the same columns.
@pytest.mark.parametrize('ray_serial', [False, True], ids=['ray-per-lane', 'ray-serial'])
def test_c3_caii_columns_per_class_path(hip_lib, oracle_lib, ray_serial):
    """C3: CaII, 64 columns (12 full column groups + one of 4), ray-dependent device-built profiles; tile classes 0, 1 (one
    line), 2 (H & K overlap)"""
    table = _run_pair(hip_lib, oracle_lib, 'falc_ca.npz', 64, 1234, 1e-12, [(0, 0, 0, 0), (1, 1, 0, 0), (2, 2, 0, 1)],      # H & K share their lower level: relation 1
                      sweep_policy='ray-serial' if ray_serial else 'ray-per-lane', options='rs_max_npt=2')
    assert sum(t for t, _ in table.values()) == 25          # 25 tiles for FALC CaII
    assert all(v == ray_serial for v in class_table.ray_serial.values()), class_table.ray_serial


@pytest.mark.parametrize('ray_serial', [False, True], ids=['ray-per-lane', 'ray-serial'])
def test_c4_cah_columns_linked_continua(hip_lib, oracle_lib, ray_serial):
    """C4: Ca+H, 41 columns.  Every hydrogen line tile carries linked continua: classes (1 line) and (2 lines), each with and
    without linked continua, plus the continuum-only tiles"""
    table = _run_pair(hip_lib, oracle_lib, 'falc_cah.npz', 41, 4321, 3e-11, [(0, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 0), (2, 2, 0, 1), (2, 2, 1, 1)],
                      sweep_policy='ray-serial' if ray_serial else 'ray-per-lane', options='rs_max_npt=2')
    assert -1 not in [k[0] for k in table]                  # no tile falls back to the generic instance
    assert max(k[0] for k in table) == 2                    # no continuum goes through the sweep
    assert all(v == ray_serial for v in class_table.ray_serial.values()), class_table.ray_serial


def test_c4_cah_columns_three_and_four_slot_instances(hip_lib, oracle_lib):
    """the same columns with the linking switched off: hydrogen's continua become per-ray slots again, and
    lsx_sweep_kernel<3, {1,2}, 5, false> and <4, {1,2}, 5, false> meet the oracle"""
    table = _run_pair(hip_lib, oracle_lib, 'falc_cah.npz', 40, 4321, 3e-11,
                      [(1, 1, 0, 0), (2, 1, 0, 0), (2, 2, 0, 1), (3, 1, 0, 0), (3, 2, 0, 0), (4, 1, 0, 0), (4, 2, 0, 0)], options='linked=0')
    assert -1 not in [k[0] for k in table]


# The reference's own larger atoms through the production launch path (round 5; fixtures generated from the reference:
# tests/golden/make_golden.py, gen_falc_multilevel; the single-column calls meet the reference's golden vectors in
# tests/test_hip_parity.py).  Classes as the product's planner files them (tests/test_instance_ledger.py walks the same plans on
# the CPU): carbon -- 119 tiles, continua linked to one line; iron -- 78 tiles; MgII -- continua linked to one, two and THREE lines
# (class (3, 3, linked)) and tiles the generic instance takes; with the linking switched off the continua of a line's atom become
# per-ray slots: (2, 1) for carbon, (4, 1) for MgII -- per-ray-continuum instances planned by a REFERENCE problem.
MULTILEVEL = {
    'falc_c.npz': ([(0, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 0), (2, 2, 0, 0), (2, 2, 0, 1)], [(2, 1, 0, 0), (-1, 0, 0, 0)]),
    'falc_fe.npz': ([(0, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 0), (2, 2, 0, 0), (2, 2, 0, 1)], [(2, 2, 0, 0), (2, 2, 0, 1)]),
    'falc_mg.npz': ([(0, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 0), (2, 2, 0, 0), (2, 2, 1, 0), (2, 2, 1, 1), (3, 3, 1, 0), (-1, 0, 1, 0)],
                    [(4, 1, 0, 0), (-1, 0, 0, 0)]),
    # all five model atoms active (gen_falc_all): 334 tiles; two lines of DIFFERENT atoms in a tile (relation 2) planned by a reference problem
    'falc_all.npz': ([(0, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 0), (2, 2, 0, 0), (2, 2, 0, 1), (2, 2, 0, 2), (2, 2, 1, 0), (2, 2, 1, 1), (3, 3, 1, 0), (-1, 0, 1, 0)],
                     [(2, 1, 0, 0), (3, 1, 0, 0), (4, 1, 0, 0), (-1, 0, 0, 0)]),
}


@pytest.mark.parametrize('mode', ['ray-per-lane', 'ray-serial', 'unlinked'])
@pytest.mark.parametrize('name', sorted(MULTILEVEL))
def test_multilevel_reference_atoms_per_class_path(hip_lib, oracle_lib, name, mode):
    linked, unlinked = MULTILEVEL[name]
    # All five atoms from the LTE start: the eighth iteration of these columns is ill-conditioned -- J passes through zero at one
    # wavelength of column 24 (dJ = 121) and the ORACLE'S OWN runs with exp() moved by +-1 ulp spread by 2.3e-5 in dJ and 1.8e-5 in J
    # there (HIP: 5.2e-5; profiles/r05/diag_all_atoms_columns.txt, profiles/diag_all_atoms.py).  Seven iterations are compared: four
    # statistical equilibria of five atoms each.
    iters = 7 if name == 'falc_all.npz' else 8
    if mode == 'unlinked':
        table = _run_pair(hip_lib, oracle_lib, name, 33, 2468, 3e-11, unlinked, iters=iters, options='linked=0')
    else:
        table = _run_pair(hip_lib, oracle_lib, name, 36, 2468, 3e-11, linked, iters=iters, sweep_policy=mode)
        # the classes with at most two per-ray slots run the mapping that was asked for, the others one ray per lane -- and so does a
        # class with linked continua if one of its tiles needs the row-mapped epilogue (more than six continua of an atom at a
        # wavelength, as carbon's and magnesium's have: the ray-serial instances leave the linked corrections to the column-mapped one)
        for key, serial in class_table.ray_serial.items():
            want = mode == 'ray-serial' and 0 <= key[0] <= 2
            assert serial == want or (want and key[2] == 1 and not serial), (key, serial)
        if mode == 'ray-serial':
            assert any(class_table.ray_serial.values())


def test_single_column_reaches_the_fused_kernel(hip_lib):
    """the counterpart: fewer than 32 columns take ONE fused launch (what the golden single-column tests exercise)"""
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'))
    e = Engine(prob, 1, lib=hip_lib)
    e.set_columns(0, base)
    e.formal_sol_gamma()
    table, fused = class_table(hip_lib, e)
    assert fused == 1 and all(launches == 0 for _, launches in table.values())
    e.close()


# (falc_mg / falc_all: atoms of 11 and 15 levels -- the many-column epilogue is k_gamma_finish_split there, a thread per block of Gamma's columns)
@pytest.mark.parametrize('name,ncol', [('falc_ca.npz', 1), ('falc_cah.npz', 3), ('falc_mg.npz', 2), ('falc_all.npz', 2)])
def test_small_batch_gamma_epilogue_gives_the_bits_of_the_many_column_one(hip_lib, name, ncol):
    """fewer than 32 columns: the Gamma epilogue runs one wavefront per (column, depth) with one entry per lane
    (k_gamma_finish_small); it sums every entry in the order the many-column kernel does, so Gamma, the monitors and the
    populations after five iterations are the same bits.  A second stat_equil on the same Gamma (the epilogue has zeroed the
    per-column maxima only once: the call has to do it itself) solves the same system again: it reports no change, not the
    first call's."""
    prob, base, raw = fixtures.load_problem_npz(golden(name), phi_compact=False)
    blk, (aD, vB, vlos) = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=11, vlos_sigma=1.0e3)
    out = []
    # the small-batch kernel, the many-column one (a thread per column of Gamma, round 5) and the many-column one with a thread's whole
    # matrix in LDS (finish_lds=1: the default before; atoms of more than six levels take the per-column kernel there too)
    for opts in ('finish_big=0', 'finish_big=1', 'finish_big=1,finish_lds=1'):
        e = Engine(prob, ncol, lib=hip_lib, options=opts)
        e.set_columns(0, blk)
        e.set_line_profiles(0, aD, vB, vlos)
        mon = []
        for it in range(5):
            mon.append(e.formal_sol_gamma())
            if it >= 2:
                mon.append(e.stat_equil())
        mon.append(e.stat_equil())                                  # twice on one Gamma
        out.append((e.get(_capi.LSX_GAMMA), e.get(_capi.LSX_N), e.get(_capi.LSX_DJ_COL), e.get(_capi.LSX_DPOPS_COL), mon))
        e.close()
    for other in out[1:]:
        for a, b in zip(out[0][:4], other[:4]):
            assert np.array_equal(a, b)
        assert out[0][4] == other[4]
    assert out[0][4][-1] == 0.0 < out[0][4][-2]


def test_profiles_must_be_set_before_a_formal_solution(hip_lib):
    """lsx_set_columns with phi == NULL leaves the profiles to lsx_set_line_profiles; a formal solution in between is
    refused instead of reading uninitialised memory"""
    prob, base, raw = fixtures.load_problem_npz(golden('falc_ca.npz'), phi_compact=False)
    blk, (aD, vB, vlos) = synth.perturbed_columns(prob, base, raw, ncol=2, seed=5)
    e = Engine(prob, 2, lib=hip_lib)
    e.set_columns(0, blk)
    with pytest.raises(_capi.LsxError, match='no line profiles'):
        e.formal_sol_gamma()
    e.set_line_profiles(0, aD[:1], vB[:1], vlos[:1])
    with pytest.raises(_capi.LsxError, match='column 1'):
        e.formal_sol_gamma()
    e.set_line_profiles(1, aD[1:], vB[1:], vlos[1:])
    assert e.formal_sol_gamma() == 1.0
    e.close()


@pytest.mark.parametrize('mode', ['default', 'mixed', 'ray-per-lane'])
@pytest.mark.parametrize('name,ncol,nuniq', [('falc_ca.npz', 1000, 8), ('falc_cah.npz', 1250, 10)])
def test_full_size_batches_by_size_independent_properties(hip_lib, oracle_lib, name, ncol, nuniq, mode):
    """BASELINE sizes (C3: 1000 CaII columns, C4: one GPU's 1250 Ca+H columns), where the oracle would take minutes:
    columns are independent 1-D problems, so a batch built from `nuniq` distinct columns repeated in a scrambled order must
    give every copy the bits its original gets in a small batch -- whatever its position, its neighbours, its place inside
    a five-column wavefront or the size of the grid; the small batch itself is checked against the oracle.
    default (what a context of this size runs: the ray-serial kernel for every class, all of which have at most two per-ray
    slots here) and mixed (options='rs_max_npt=1': ray-serial for tiles with at most one per-ray slot, one ray per lane for the
    two-line tiles, side by side in one call): the small batch is 37 columns (seven column groups and one of two) forced
    onto the same kernels; ray-per-lane (sweep_policy='ray-per-lane'): the small batch takes the fused launch, the big one the per-class
    launches.  Also: a frozen column keeps its state bit for bit while its neighbours
    iterate (lsx_set_active_columns)."""
    ray_serial = mode != 'ray-per-lane'
    prob, base, raw = fixtures.load_problem_npz(golden(name), phi_compact=False)
    blk, prof = synth.perturbed_columns(prob, base, raw, ncol=nuniq, seed=77, vlos_sigma=2.0e3)
    rng = np.random.default_rng(5)
    src = np.concatenate([np.arange(nuniq), rng.integers(0, nuniq, ncol - nuniq)])
    nsmall = 37 if ray_serial else nuniq
    ssrc = src[:nsmall]
    pick = lambda idx: (type(blk).concatenate([blk.slice(int(q), int(q) + 1) for q in idx]), tuple(p[idx] for p in prof))
    # explicit arguments, not the environment: the small batch is made for the BIG problem's column count (policy_columns: the
    # kernel choice belongs to the problem), `mixed` leaves the two-slot tiles to one ray per lane, `ray-per-lane` pins that mapping
    kw = dict(options='rs_max_npt=1' if mode == 'mixed' else None, sweep_policy='auto' if ray_serial else 'ray-per-lane')
    small = Engine(prob, nsmall, lib=hip_lib, policy_columns=ncol, **kw)
    big = Engine(prob, ncol, lib=hip_lib, **kw)
    synth.load_columns(small, *pick(ssrc))
    synth.load_columns(big, *pick(src))
    ora = Engine(prob, nuniq, lib=oracle_lib)
    synth.load_columns(ora, blk, prof)
    oracle_lib.dll.lsx_oracle_set_threads(ora._h, 8)
    for it in range(1, 7):
        dJs, dJb = small.formal_sol_gamma(), big.formal_sol_gamma()
        ora.formal_sol_gamma()
        assert dJs == dJb
        if it > 3:
            assert small.stat_equil() == big.stat_equil()
            ora.stat_equil()
    first = np.array([int(np.nonzero(ssrc == q)[0][0]) for q in range(nuniq)])      # where each original sits in the small batch
    for what in (_capi.LSX_J, _capi.LSX_I, _capi.LSX_N, _capi.LSX_GAMMA, _capi.LSX_DJ_COL, _capi.LSX_DPOPS_COL):
        a, b = small.get(what), big.get(what)
        assert np.array_equal(a, a[first][ssrc]), what             # every copy inside the small batch
        assert np.array_equal(b, a[first][src]), what              # bit for bit, every copy of the big one
    tol = 1e-8
    assert relerr(small.get(_capi.LSX_N)[first], ora.get(_capi.LSX_N)) < tol and relerr(small.get(_capi.LSX_I)[first], ora.get(_capi.LSX_I)) < tol
    table, fused = class_table(hip_lib, big)
    assert fused == 0 and all(launches == 6 for _, launches in table.values())
    expect = {key: (ray_serial and (mode == 'default' or key[0] <= 1)) for key in table}
    assert class_table.ray_serial == expect, class_table.ray_serial
    if ray_serial:
        table, fused = class_table(hip_lib, small)
        assert fused == 0 and class_table.ray_serial == expect      # the small batch ran the same kernels
    else:
        assert class_table(hip_lib, small)[1] == 6                 # the small batch took the fused kernel
    # frozen columns: every second column is frozen for two more iterations
    mask = (np.arange(ncol) % 2 == 0)
    n0, J0 = big.get(_capi.LSX_N), big.get(_capi.LSX_J)
    big.set_active_columns(mask)
    for _ in range(2):
        big.formal_sol_gamma()
        big.stat_equil()
    n1, J1 = big.get(_capi.LSX_N), big.get(_capi.LSX_J)
    assert np.array_equal(n1[~mask], n0[~mask]) and np.array_equal(J1[~mask], J0[~mask])
    assert not np.array_equal(n1[mask], n0[mask])
    big.set_active_columns(None)
    small.close(); big.close(); ora.close()


@pytest.mark.parametrize('name,ncol', [('falc_ca.npz', 1), ('falc_cah.npz', 3), ('falc_cah.npz', 31)])
def test_fused_launch_runs_the_fast_continuum_work_itself_with_the_same_bits(hip_lib, name, ncol):
    """fewer than 32 columns: the workgroup of a tile with fast continua runs the tile's pre-pass itself, before its own sweep, inside
    the ONE fused launch (lsx_sweep.hip, lsx_fast.h) instead of a launch in front of it (LSX_NO_FUSED_FAST=1 restores that; with
    LSX_FUSED_EPILOGUE=1 the Gamma epilogue runs inside as well): the same device functions, the same bits -- J, I, Gamma, populations
    and monitors after five iterations, Ca+H with linked continua (one and two lines per tile) included"""
    prob, base, raw = fixtures.load_problem_npz(golden(name), phi_compact=False)
    blk, (aD, vB, vlos) = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=21, vlos_sigma=1.5e3)
    out = []
    for leg in ('default', 'separate', 'epilogue-inside'):      # default: the pre-pass inside the fused launch, the epilogue a launch of its own
        # (epilogue-inside was measured slower: the epilogue then extends the longest workgroups)
        e = Engine(prob, ncol, lib=hip_lib, options={'default': None, 'separate': 'fused_fast=0', 'epilogue-inside': 'fused_epilogue=1'}[leg])
        e.set_columns(0, blk)
        e.set_line_profiles(0, aD, vB, vlos)
        mon = []
        for it in range(5):
            mon.append(e.formal_sol_gamma())
            if it >= 2:
                mon.append(e.stat_equil())
        out.append(([e.get(w) for w in (_capi.LSX_J, _capi.LSX_I, _capi.LSX_GAMMA, _capi.LSX_N, _capi.LSX_DJ_COL)], mon))
        assert class_table(hip_lib, e)[1] == 5          # five fused launches either way
        e.close()
    for other in out[1:]:
        for a, b in zip(out[0][0], other[0]):
            assert np.array_equal(a, b)
        assert out[0][1] == other[1]


@pytest.mark.gpu
@pytest.mark.parametrize('name,ncol,policy', [('falc_ca.npz', 1, 'auto'), ('falc_cah.npz', 40, 'auto'), ('falc_cah.npz', 36, 'ray-serial')])
def test_formal_solution_replayed_as_a_captured_graph_gives_the_same_bits(hip_lib, name, ncol, policy):
    """LSX_GRAPH=1 (a measurement switch): the launches of a formal solution -- fork, the classes' chains on their streams,
    join, Gamma epilogue -- are captured once per (J buffer parity, result buffers, epilogue flavour, rule, mapping, mask) and replayed
    as a HIP graph.  The same kernels with the same arguments: J, I, Gamma, populations and monitors bit for bit, through the
    synchronous calls, the enqueue-only sequence FS; SE; FS; sync, a frozen column and a change of rule."""
    prob, base, raw = fixtures.load_problem_npz(golden(name), phi_compact=False)
    blk, prof = synth.perturbed_columns(prob, base, raw, ncol=ncol, seed=31, vlos_sigma=1.5e3)
    out = []
    for leg in ('eager', 'graph'):
        # (ray-serial: the folded instances and the operand table, which is rebuilt OUTSIDE the captured launches whenever the
        # populations have changed -- behind the statistical equilibrium's read-back or in front of the next formal solution)
        e = Engine(prob, ncol, lib=hip_lib, options='graph=%d' % (leg == 'graph'), sweep_policy=policy)
        synth.load_columns(e, blk, prof)
        mon = []
        for it in range(5):
            mon.append(e.formal_sol_gamma())
            if it >= 2:
                mon.append(e.stat_equil())
        e.formal_sol_gamma_async(); e.stat_equil_async(); e.formal_sol_gamma_async()
        mon.append(e.sync())
        if ncol > 1:
            e.set_active_columns(np.arange(ncol) % 2 == 0)
            mon.append(e.formal_sol_gamma()); mon.append(e.stat_equil())
            e.set_active_columns(None)
        e.set_formal_solver('parabolic')
        mon.append(e.formal_sol_gamma())
        e.set_formal_solver('linear')
        mon.append(e.formal_sol_gamma())
        out.append((mon, [e.get(w) for w in (_capi.LSX_J, _capi.LSX_I, _capi.LSX_N, _capi.LSX_GAMMA, _capi.LSX_DJ_COL, _capi.LSX_DPOPS_COL)]))
        e.close()
    (m0, a0), (m1, a1) = out
    assert repr(m0) == repr(m1)
    for x, y in zip(a0, a1):
        assert np.array_equal(x, y)
