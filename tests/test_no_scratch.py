"""No kernel of the product uses scratch memory.  A spilled vector register is a round trip to memory in
the middle of a recurrence; round 4 had removed them from the instances the BASELINE configurations plan and left 4 - 120 bytes per
lane in the off-headline ones (two and three per-ray slots with linked continua, four slots, tile widths other than twelve).
The Makefile leaves the compiler's per-kernel resource report of every device translation unit beside its object
(build/<unit>.ru.log, -Rpass-analysis=kernel-resource-usage); this test reads them: ScratchSize 0, no spilled vector register, no
dynamic stack -- for EVERY kernel in the library, i.e. for every instance the planner can dispatch (tests/test_instance_ledger.py
walks the instance lists) and everything around them.  (Scalar registers spilled into vector lanes are not memory traffic.)"""
import glob
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, 'lightspinner_amd', 'csrc')
UNITS = ('lsx_sweep', 'lsx_sweep_rs', 'lsx_sweep_rs_par', 'lsx_hip', 'lsx_setup')


def _reports():
    logs = [os.path.join(CSRC, 'build', u + '.ru.log') for u in UNITS]
    if not all(os.path.exists(f) for f in logs):
        if shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'):
            pytest.skip('no hipcc and no resource reports')
        subprocess.check_call(['make', '-s', '-j', '8', '-C', CSRC])
    out = {}
    for f in logs:
        txt = open(f).read()
        for blk in re.split(r'remark: Function Name: ', txt)[1:]:
            name = blk.split()[0]
            get = lambda key: re.search(re.escape(key) + r': (\S+)', blk).group(1)
            out[(os.path.basename(f), name)] = dict(scratch=int(get('ScratchSize [bytes/lane]')), vspill=int(get('VGPRs Spill')),
                                                    dynstack=get('Dynamic Stack'), vgpr=int(get('VGPRs')), agpr=int(get('AGPRs')),
                                                    occupancy=int(get('Occupancy [waves/SIMD]')))
    return out


def test_no_kernel_uses_scratch_or_spills_a_vector_register():
    rep = _reports()
    kernels = {k: v for k, v in rep.items()}
    assert len(kernels) > 150, len(kernels)                    # every instance of both sweeps, the fast-continuum kernels, the set-up chain
    bad = {k: v for k, v in kernels.items() if v['scratch'] != 0 or v['vspill'] != 0 or v['dynstack'] != 'False'}
    assert not bad, bad
    # the register budgets the design relies on: the ray-serial instances fit two waves per SIMD, the continuum-tile
    # instance three -- folded or not
    rs = {n: v for (f, n), v in kernels.items() if f == 'lsx_sweep_rs.ru.log' and 'lsx_sweep_rs_kernel' in n}
    assert len(rs) >= 16 and all(v['occupancy'] >= 2 for v in rs.values())
    assert all(v['occupancy'] >= 3 for n, v in rs.items() if 'ILi0ELi0E' in n)
    # the parabolic ray-serial instances: one wave per SIMD with accumulation registers as spill space -- still no scratch
    rsp = {n: v for (f, n), v in kernels.items() if f == 'lsx_sweep_rs_par.ru.log' and 'lsx_sweep_rs_kernel' in n}
    assert len(rsp) == 5 and all(v['scratch'] == 0 for v in rsp.values())


def test_the_reports_belong_to_the_sources():
    """the reports are as new as the objects they were written with (a stale report would make the test above vacuous)"""
    for u in UNITS:
        log, obj = os.path.join(CSRC, 'build', u + '.ru.log'), os.path.join(CSRC, 'build', u + '.o')
        if not (os.path.exists(log) and os.path.exists(obj)):
            pytest.skip('not built here')
        src = os.path.join(CSRC, ('lsx_sweep_rs' if u.startswith('lsx_sweep_rs') else u) + '.hip')
        assert os.path.getmtime(log) >= os.path.getmtime(src) - 1.0, u
        # ... and belongs to the object beside it: the compiler writes both in one run (a cached object with a deleted or older
        # report would let the test above pass on nothing)
        assert abs(os.path.getmtime(log) - os.path.getmtime(obj)) < 120.0, (u, 'the report is not from the compile that made the object: make clean && make')
        assert 'Function Name' in open(log).read(), u
