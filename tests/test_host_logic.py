"""CPU tests of the host side: C-ABI export surface, the drop-in Context (packing, aliasing,
warm start, error behaviour) with the oracle bound as a checker, sharding helpers, synthetic
columns.  No GPU needed."""
import os
import re

import numpy as np
import pytest

from conftest import golden, relerr, gamma_err, ROOT
from helpers import build_fakes, build_data_fakes
import context_cases
import grid_cases
from lightspinner_amd import _capi, fixtures, synth, drivers
from lightspinner_amd.parallel import shard_columns
from lightspinner_amd.rh_method import Context


def _declared_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'lsx.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(lsx_[a-z0-9_]+)\s*\(', hdr)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_capi.REQUIRED_SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    """the product .so must be built in-tree and export the whole ABI (no compute call here)"""
    import ctypes
    path = _capi.hip_library_path()
    assert os.path.exists(path), 'run __graft_entry__.build() first'
    dll = ctypes.CDLL(path)
    for s in _declared_symbols():
        assert hasattr(dll, s), s
    lib = _capi.LsxLibrary(path)
    assert lib.backend == 'hip-gfx950'


def test_oracle_exports_the_same_abi(oracle_lib):
    for s in _declared_symbols():
        assert hasattr(oracle_lib.dll, s), s


def test_hip_engine_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    prob, block, _ = fixtures.load_problem_npz(golden('falc_ca.npz'))
    from lightspinner_amd import Engine
    with pytest.raises(_capi.LsxError):
        Engine(prob, 1)     # default library = HIP; there is no CPU fallback


def test_context_dropin_matches_reference_golden(oracle_lib):
    context_cases.context_dropin_matches_reference_golden(oracle_lib)


def test_context_warm_start_and_host_edits(oracle_lib):
    context_cases.context_warm_start_and_host_edits(oracle_lib)


def test_context_lazy_readback_keeps_the_reference_semantics(oracle_lib):
    context_cases.context_lazy_readback_keeps_the_reference_semantics(oracle_lib)


def test_context_lookahead_gives_the_plain_sequence_bit_for_bit(oracle_lib):
    context_cases.context_lookahead_gives_the_plain_sequence_bit_for_bit(oracle_lib)


def test_context_two_active_atoms_order_and_shapes(oracle_lib):
    context_cases.context_two_active_atoms_order_and_shapes(oracle_lib)


@pytest.mark.parametrize('name', ['falc_cah.npz', 'falc_ca.npz', 'falc_ca_vlos.npz'])
def test_context_native_setup_chain(oracle_lib, name):
    context_cases.context_native_setup_chain(oracle_lib, name)


def test_context_methods_setup(oracle_lib):
    context_cases.context_methods_setup_is_still_the_reference_interface(oracle_lib)


def test_piecewise_linear_1d_dropin(oracle_lib):
    context_cases.piecewise_linear_1d_dropin(oracle_lib)


def test_golden_w2_and_piecewise_1d_impl_through_the_abi(oracle_lib):
    context_cases.golden_w2_and_piecewise_1d_impl(oracle_lib)


def test_dead_level_nan_is_dropped_from_dpops(oracle_lib):
    context_cases.dead_level_nan_is_dropped_from_dpops(oracle_lib)


def test_wavelength_grid_bit_exact_on_the_reference(oracle_lib):
    grid_cases.reference_grid_bit_exact(oracle_lib)


def test_line_grids_and_continuum_alpha(oracle_lib):
    grid_cases.line_grids_and_continuum_alpha(oracle_lib)


def test_wavelength_grid_random_and_edge_cases(oracle_lib):
    grid_cases.random_and_edge_cases(oracle_lib)


def test_spectrum_configuration_feeds_context(oracle_lib):
    grid_cases.spectrum_configuration_feeds_context(oracle_lib, oracle_lib)


def test_shard_columns_partitions_exactly():
    for total in (0, 1, 7, 8, 165, 1000, 10000):
        for world in (1, 2, 3, 8):
            parts = [shard_columns(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(n for _, n in parts) == total
            for (f0, n0), (f1, _) in zip(parts, parts[1:]):
                assert f1 == f0 + n0
            assert max(n for _, n in parts) - min(n for _, n in parts) <= 1


def test_synthetic_columns_are_deterministic_per_absolute_index():
    prob, block, raw = fixtures.load_problem_npz(golden('falc_ca.npz'), phi_compact=False)
    whole, wp = synth.perturbed_columns(prob, block, raw, ncol=5, seed=7)
    part, pp = synth.perturbed_columns(prob, block, raw, ncol=2, seed=7, first=3)
    for k in ('bg_chi', 'n', 'C'):
        assert np.array_equal(getattr(whole, k)[3:5], getattr(part, k))
    for w, q in zip(wp, pp):
        assert np.array_equal(w[3:5], q)
    assert np.array_equal(whole.bg_chi[0], block.bg_chi[0]) and np.all(wp[2][0] == 0.0)   # column 0 is FALC itself
    assert not np.array_equal(wp[2][1], wp[2][2])                    # vlos differs from column to column
    same, none = synth.perturbed_columns(prob, block, raw, ncol=3, seed=7, vlos_sigma=0.0)
    assert none is None and np.array_equal(same.phi[2], block.phi[0])    # no velocity: the base column's profiles


def test_response_function_formula():
    """response_fn.py:59-67"""
    rng = np.random.default_rng(0)
    Ip, Im = rng.random((82, 10, 5)), rng.random((82, 10, 5))
    Ib = rng.random((10, 5)) + 1
    rf = drivers.response_function(Ip, Im, Ib)
    assert rf.shape == (10, 82)
    assert rf[3, 40] == pytest.approx((Ip[40, 3, 4] - Im[40, 3, 4]) / Ib[3, 4])


def test_columns_without_profiles_and_device_profile_inputs():
    """ColumnBlock may leave phi / wphi to lsx_set_line_profiles; synth hands over the profile inputs instead"""
    prob, block, raw = fixtures.load_problem_npz(golden('falc_ca.npz'), phi_compact=False)
    blk, (aD, vB, vlos) = synth.perturbed_columns(prob, block, raw, ncol=4, seed=7)
    assert blk.phi is None and blk.wphi is None and blk.ncol == 4
    assert aD.shape == (4, prob.Nlines, prob.Nspace) and vB.shape == (4, prob.Natoms, prob.Nspace) and vlos.shape == (4, prob.Nspace)
    assert np.all(vlos[0] == 0.0) and np.any(vlos[1] != 0.0)            # column 0 is the unperturbed FALC column
    host, _ = synth.perturbed_columns(prob, block, raw, ncol=4, seed=7, vlos_sigma=0.0)   # same ensemble, no velocity
    for k in ('bg_chi', 'n', 'C', 'nStar'):
        assert np.array_equal(getattr(blk, k), getattr(host, k))
    part = blk.slice(1, 3)
    assert part.phi is None and part.ncol == 2
    s = part.to_c()
    assert not s.phi and not s.wphi                                       # NULL pointers in lsx_columns
    both = type(blk).concatenate([part, part])
    assert both.phi is None and both.ncol == 4
    bad = blk.slice(0, 1)
    bad.wphi = np.zeros((1, prob.Nlines, prob.Nspace))
    with pytest.raises(ValueError):
        bad.validate(prob)


def test_header_is_plain_c_and_cxx():
    """include/lsx.h is the drop-in boundary: it must compile as C99 and as C++11 on its own (no torch, no HIP types)"""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, 'include', 'lsx.h')
    for cc, flags in (('gcc', ['-std=c99', '-pedantic', '-x', 'c']), ('g++', ['-std=c++11', '-x', 'c++'])):
        if shutil.which(cc) is None:
            pytest.skip(cc + ' not installed')
        r = subprocess.run([cc, '-Wall', '-Werror', '-fsyntax-only'] + flags + [hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    text = open(hdr).read()
    assert '#include <torch' not in text and '#include <hip' not in text and 'at::Tensor' not in text      # a stream is passed as void*


def test_import_asks_for_eight_hardware_queues_unless_the_caller_has_chosen():
    """lightspinner_amd/__init__.py: GPU_MAX_HW_QUEUES=8 (six tile classes on streams of their own) is a default, not an
    override -- checked in fresh interpreters, without touching a GPU"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, lightspinner_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    for preset, expect in ((None, '8'), ('4', '4')):
        env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
        if preset is not None:
            env['GPU_MAX_HW_QUEUES'] = preset
        out = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == expect
