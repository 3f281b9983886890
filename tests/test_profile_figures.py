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
