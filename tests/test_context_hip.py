"""The drop-in boundary on the PRODUCT library (GPU): lightspinner_amd.rh_method.Context and
lightspinner_amd.formal_solver.piecewise_linear_1d with their default backend (lib=None -> liblsx_hip.so), on the
reference-shaped stand-in objects of tests/helpers.py, against the reference's golden vectors; the reference's unit
vectors for w2 / piecewise_1d_impl through the HIP C ABI.  Same bodies as the CPU run (tests/context_cases.py)."""
import numpy as np
import pytest

import context_cases
import grid_cases

pytestmark = pytest.mark.gpu


def test_context_on_hip_matches_reference_golden(hip_lib):
    context_cases.context_dropin_matches_reference_golden(None)        # None: the product default, i.e. HIP


def test_context_on_hip_warm_start_and_host_edits(hip_lib):
    context_cases.context_warm_start_and_host_edits(None)


def test_context_on_hip_lazy_readback(hip_lib):
    context_cases.context_lazy_readback_keeps_the_reference_semantics(None)


def test_context_on_hip_lookahead_gives_the_plain_sequence(hip_lib):
    context_cases.context_lookahead_gives_the_plain_sequence_bit_for_bit(None)


def test_context_on_hip_two_active_atoms(hip_lib):
    context_cases.context_two_active_atoms_order_and_shapes(None)


def test_piecewise_linear_1d_dropin_on_hip(hip_lib):
    context_cases.piecewise_linear_1d_dropin(None)


@pytest.mark.parametrize('name', ['falc_cah.npz', 'falc_ca.npz', 'falc_ca_vlos.npz'])
def test_context_native_setup_chain_on_hip(name):
    context_cases.context_native_setup_chain(None, name)


def test_context_methods_setup_on_hip():
    context_cases.context_methods_setup_is_still_the_reference_interface(None)


def test_golden_w2_and_piecewise_1d_impl_on_hip(hip_lib):
    context_cases.golden_w2_and_piecewise_1d_impl(hip_lib)


def test_dead_level_nan_is_dropped_from_dpops_on_hip(hip_lib, oracle_lib):
    a = context_cases.dead_level_nan_is_dropped_from_dpops(hip_lib)
    b = context_cases.dead_level_nan_is_dropped_from_dpops(oracle_lib)
    assert a == pytest.approx(b, rel=1e-6)


def test_wavelength_grid_on_the_product_library(hip_lib):
    """lsx_grid.cpp (host C++ inside liblsx_hip.so): the same bodies as the oracle's CPU run"""
    grid_cases.reference_grid_bit_exact(hip_lib)
    grid_cases.line_grids_and_continuum_alpha(hip_lib)
    grid_cases.random_and_edge_cases(hip_lib)


def test_spectrum_configuration_feeds_context_on_hip(hip_lib):
    grid_cases.spectrum_configuration_feeds_context(None, None)
