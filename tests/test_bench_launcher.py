"""`python bench.py --gpus N` starts its own ranks (the 1/2/4/8-GPU runs).  CPU checks of the launcher: the
parent process touches no GPU (it must not, a process that has initialised HIP may not start the ranks), the ranks
rendezvous over 127.0.0.1 and exactly one JSON line reaches stdout."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_self_launch_parent_stays_off_the_gpu():
    code = r'''
import json, os, sys
sys.path.insert(0, %r)
import bench, subprocess
seen = {}
class FakeProc:
    def __init__(self, cmd, env=None, stdout=None, text=None):
        seen['cmd'], seen['env'] = cmd, env
        self.stdout = iter(['rank chatter\n', '{"metric": "m", "n_gpus": 4}\n'])
    def wait(self): return 0
subprocess.Popen = FakeProc
sys.argv = ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2']
os.environ.pop('WORLD_SIZE', None)
bench.main()
maps = open('/proc/self/maps').read()
print(json.dumps(dict(cmd=seen['cmd'], torch='torch' in sys.modules, hip='liblsx_hip' in maps or 'libamdhip64' in maps,
                      ipc=seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY'))))
''' % ROOT
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert json.loads(lines[0]) == {'metric': 'm', 'n_gpus': 4}         # rank 0's line, passed through
    info = json.loads(lines[1])
    assert not info['torch'] and not info['hip']                        # the parent never imported torch / loaded HIP
    assert info['ipc'] == '0'
    cmd = info['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-7].endswith('bench.py') and cmd[-6:] == ['--gpus', '4', '--steps', '7', '--warmup', '2']


def test_self_launch_two_ranks_dry_run():
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--dry-run'],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['steps'] == 3 and r['max_rank_seen'] == 1 and r['dry_run'] is True


def test_gpus_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and 'WORLD_SIZE=1' in out.stderr


import math

import pytest

from conftest import REHEARSALS, rehearsal_output


@pytest.mark.gpu
@pytest.mark.parametrize('which', list(REHEARSALS))
def test_two_rank_rehearsal_on_one_gpu(which):
    """SURVEY 8e while no multi-GPU node is at hand: the N > 1 flow of bench.py end to end on ONE GPU -- the launcher parent (which
    touches no GPU) starts two ranks, both on GPU 0, exchanging over gloo (LSX_BENCH_REHEARSE=1; RCCL refuses two ranks on one
    device); rank 0 prints ONE JSON line, labelled as a rehearsal.  C3: 200 columns per rank, the all-reduce(MAX) of the monitors
    per iteration; C5: the 164 perturbed columns of the response function sharded 82 / 82, per-column convergence, all_done = AND over
    ranks, the intensities gathered -- and the same rf as the reference at its three depths.
    (The two commands are started by tests/conftest.py when collection ends, before this process initialises the GPU.)"""
    rc, out, err = rehearsal_output(which)
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r['rehearsal'] is True and r['backend'] == 'gloo' and r['n_gpus'] == 2
    assert math.isfinite(r['value']) and r['value'] > 0 and math.isfinite(r['ms_per_step'])
    if which.startswith('c5'):
        assert r['response_function']['max_abs_err_vs_reference_rf_over_max'] < 1e-5
        assert r['scaling'] == 'strong'
    else:
        assert r['steps'] == 5 and r['config']['columns_total'] == 400 and math.isfinite(r['last_dJ'])
        assert r['config']['sweep_policy'] == 'ray-serial'        # decided for the 400 columns of the job, not for a rank's 200
        # the scaling record proves itself: what every rank saw (round 5) -- both ranks, the world size they were in, their backend,
        # their own time, the mapping they ran and one options signature for the whole job
        sr = r['scaling_record']
        assert sr['world_seen'] == 2 and sr['backend'] == 'gloo' and [q['rank'] for q in sr['per_rank']] == [0, 1]
        for q in sr['per_rank']:
            assert q['world_seen'] == 2 and q['columns'] == 200 and q['sweep_policy'] == 'ray-serial' and math.isfinite(q['ms_per_step'])
        assert len({q['options_signature'] for q in sr['per_rank']}) == 1
        assert max(q['ms_per_step'] for q in sr['per_rank']) == pytest.approx(r['ms_per_step'], rel=1e-9)
        assert 'mapping=ray-serial' in r['config']['runtime_env']['effective_options']
