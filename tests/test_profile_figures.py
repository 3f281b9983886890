"""The counter figures bench.py quotes (profiles/pmc_figures.json: HBM traffic and VALU instructions per column, from a
builder-run rocprofv3 collection) carry the hash of the kernel sources they were measured on.  bench.py withholds them
(`traffic: null`, `traffic_stale: true`) when the sources beside the library differ; at the end of a round the committed
figures must belong to the committed sources (the published traffic has to be the shipped kernels')."""
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'profiles'))
import srchash  # noqa: E402


def test_committed_figures_belong_to_the_committed_kernel_sources():
    fig = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_figures.json')))
    now = srchash.csrc_hash()
    for workload in ('c3', 'c4'):
        assert workload in fig, 'run profiles/collect.sh + profiles/summarize.py for %s' % workload
        assert fig[workload].get('csrc_hash') == now, \
            ('profiles/pmc_figures.json[%s] was collected at source hash %s, the kernel sources are at %s: re-run '
             'profiles/collect.sh and profiles/summarize.py (or bench.py publishes traffic_stale)' % (workload, fig[workload].get('csrc_hash'), now))
        assert 'valu_busy_frac' not in fig[workload]          # round 2's ratio was not a utilisation; it is not published


def test_bench_withholds_stale_figures(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    fig, src, stale = bench.profile_figures('c3')
    assert fig is not None and stale is False
    monkeypatch.setattr(srchash, 'csrc_hash', lambda: 'something else')
    fig, src, stale = bench.profile_figures('c3')
    assert stale is True


def test_figures_carry_the_ceiling_model():
    """round 6: per workload the counters of EVERY kernel of a call -- bytes, vector instructions, the alone
    time of the counter run -- and what they allow at best; bench.py derives `roofline.bound` and `roofline.ceiling` from them."""
    fig = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_figures.json')))
    for workload in ('c3', 'c4'):
        f = fig[workload]
        for key in ('hbm_bytes_per_call_per_column', 'hbm_bytes_per_call_per_column_all_kernels', 'valu_insts_per_call_per_column',
                    'valu_insts_per_call_per_column_all_kernels', 'held_clock_Hz', 'achievable_bw_Bps', 'ceiling_per_kernel',
                    'ceiling_profiled_columns'):
            assert key in f, (workload, key)
        assert f['hbm_bytes_per_call_per_column_all_kernels'] > f['hbm_bytes_per_call_per_column'] > 1e6
        assert f['valu_insts_per_call_per_column_all_kernels'] > f['valu_insts_per_call_per_column'] > 1e5
        ck = f['ceiling_per_kernel']
        assert any('lsx_sweep_rs_kernel' in k for k in ck) and any('k_gamma_finish' in k for k in ck)
        for k, v in ck.items():
            t_min = max(v['t_bytes_ms'], v['t_valu_ms'])
            assert v['bound'] == ('hbm' if v['t_bytes_ms'] >= v['t_valu_ms'] else 'valu')
            # a kernel alone cannot beat what its own bytes and instructions allow (5 %: the two clocks and the counters' granularity)
            assert v['alone_ms'] > 0 and t_min <= 1.05 * v['alone_ms'], (workload, k, v)
            # (the file keeps four decimals of a millisecond: a 17 us kernel's ratio moves by 1e-4 / 0.017)
            assert abs(v['frac_of_ceiling'] - t_min / v['alone_ms']) < max(2e-3, 2e-4 / v['alone_ms'])
        # the per-kernel bytes add up to the call's
        assert abs(sum(v['hbm_bytes_per_column'] for v in ck.values()) - f['hbm_bytes_per_call_per_column_all_kernels']) < 1e-6 * f['hbm_bytes_per_call_per_column_all_kernels']


def test_bench_line_derives_bound_and_ceiling_from_the_figures(monkeypatch):
    """bench.roofline_block on a stand-in engine (no GPU): the SURVEY 8d headline (algorithmic bytes of the call over the call's duration),
    the sweep-only figures under their own names, the renamed epilogue fields and the ceiling block"""
    sys.path.insert(0, ROOT)
    import bench
    fig = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_figures.json')))['c4']
    monkeypatch.setattr(bench, 'profile_figures', lambda wl: (fig, 'profiles/pmc_figures.json', False))

    class Dll:
        class _F:
            argtypes = restype = None

            def __call__(self, h, w):
                return {0: 6.9e6, 1: 66.0, 2: 38000.0, 3: 12.0, 4: 0.2, 5: 1e5, 6: 0.6}[w]
        lsx_hip_info = _F()

    class Lib:
        dll = Dll()

    class Eng:
        _h = None

        def time_formal_sol(self, w, r):
            return 4.5, 3.6

        def algorithmic_bytes_per_column(self):
            return 7.05e6

    class Prob:
        def work_units_per_column(self):
            return 637140
    r = bench.roofline_block(Eng(), Lib(), Prob(), 1250, 'c4', 3)
    assert abs(r['achieved'] - 7.05e6 * 1250 / 4.5e-3 / 1e9) < 1e-6 * r['achieved'] and abs(r['frac'] - r['achieved'] / 8000.0) < 1e-12
    assert abs(r['sweep_frac'] - 6.9e6 * 1250 / 3.6e-3 / 1e9 / 8000.0) < 1e-9 and r['avg_launch_ms'] == 4.5 and r['sweep_span_ms'] == 3.6
    fc = r['fs_call']
    assert fc['ms_gamma_finish'] == 0.2 and fc['ms_fast_epilogue_exposed'] == 0.6 and 'ms_epilogue_kernels' not in fc
    c = r['ceiling']
    assert c['call']['t_min_ms'] == max(c['call']['t_bytes_ms'], c['call']['t_valu_ms'])
    assert r['bound'] == c['call']['bound'] == ('hbm' if c['call']['t_bytes_ms'] >= c['call']['t_valu_ms'] else 'valu')
    assert abs(c['call']['frac_of_ceiling'] - c['call']['t_min_ms'] / 4.5) < 1e-12
    assert abs(r['traffic_over_alg'] - fig['hbm_bytes_per_call_per_column_all_kernels'] / 7.05e6) < 1e-9
    assert set(c['per_kernel_alone']) == set(fig['ceiling_per_kernel'])
