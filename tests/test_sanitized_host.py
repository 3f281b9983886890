"""SURVEY 5 "race detection / sanitizers": the product's host-only code (tile planner lsx_plan.cpp -- which sizes LDS and forms
the 32-bit offsets the kernels use -- and the wavelength-grid code lsx_grid.cpp) and the oracle run under
-fsanitize=address,undefined in the CPU suite.  GPU AddressSanitizer is not available on the pool; the kernels' operand
shapes are what the plan fixes, so the plan's invariants are verified here for the reference's problems, the toy
topologies and random transition tables (tests/san_driver.py)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


def _libasan():
    out = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True)
    p = out.stdout.strip()
    return os.path.realpath(p) if out.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


def _run(leg, lib):
    asan = _libasan()
    if asan is None:
        pytest.skip('no libasan on this host')
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:halt_on_error=1',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1', OMP_NUM_THREADS='2')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'san_driver.py'), leg, lib], capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
    tail = (out.stdout[-1500:] + '\n' + out.stderr[-3000:])
    assert out.returncode == 0, tail
    assert 'SANITIZED RUN COMPLETE' in out.stdout, tail
    assert 'AddressSanitizer' not in out.stderr and 'runtime error' not in out.stderr, tail
    return out.stdout


@pytest.mark.skipif(shutil.which('g++') is None, reason='no host compiler')
def test_plan_and_grid_under_asan_ubsan():
    csrc = os.path.join(ROOT, 'lightspinner_amd', 'csrc')
    subprocess.check_call(['make', '-s', '-C', csrc, 'asan'])
    out = _run('plan', os.path.join(csrc, 'liblsx_host_asan.so'))
    assert 'plans verified' in out


@pytest.mark.skipif(shutil.which('gcc') is None, reason='no host compiler')
def test_oracle_under_asan_ubsan():
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'asan'])
    _run('oracle', os.path.join(ROOT, 'oracle', 'liblsx_oracle_asan.so'))
