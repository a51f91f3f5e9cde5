"""The driver's contract on bench.py: one JSON line on stdout with the agreed keys, a roofline and a cpu_baseline block,
internally consistent numbers.  A short run of the real script on the GPU (secondary configurations skipped)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, chatter=False):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *flags], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    if chatter:      # (the gloo rehearsal: the backend announces its connections on stdout; RCCL does not)
        lines = [l for l in lines if l.startswith('{')]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run('--steps', '24', '--warmup', '4', '--prewarm', '32', '--no-secondary', '--cpu-samples', '4096')
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 24 and d['warmup'] == 4 and d['higher_is_better'] is True
    assert d['unit'] == 'log-likelihoods/sec' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    batch = d['config']['global_batch']
    assert abs(d['value'] - batch / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0.05 < r['frac'] < 1.0
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['kernel_ms'] * 1e-3) / 1e9) <= 1e-6 * r['achieved']
    # the kernel cannot take longer than the step around it -- the step of the loop that carried the kernel events: the
    # single-stream loop (round 6: `value` is the two-stream loop, whose launches overlap; ms_per_step may undercut the kernel)
    one = d['config'].get('ms_per_step_one_stream') or d['ms_per_step']
    assert r['kernel_ms'] <= one * 1.05
    assert d['config']['params_mode'].startswith('default') and d['config']['fp32_exact_ms'] > one
    assert r['traffic_source'] and 'profiles/' in r['traffic_source']
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and d['value'] > 10 * c['value']
    # the mean LL of N(0,1) inputs under this model: finite and the same from run to run (seeded)
    assert abs(d['config']['mean_ll'] + 1430.6) < 1.0


def test_bench_two_rank_rehearsal_on_one_device():
    """The N > 1 code path of bench.py rehearsed with two ranks on ONE device over gloo (`--share-device`; RCCL refuses two
    ranks on a device): the line names the strong-scaling shard (65536 / 2 per rank), both ranks are seen, the value is the
    whole-job aggregate.  A window whose collective cannot be captured (gloo) must leave the eager figure and say so."""
    d = _run('--gpus', '2', '--backend', 'gloo', '--share-device', '--steps', '8', '--warmup', '2', '--prewarm', '8',
             '--no-secondary', '--cpu-samples', '0', '--no-kernel-events', chatter=True)
    assert d['n_gpus'] == 2 and d['n_ranks_seen'] == 2 and d['backend'] == 'gloo' and d['scaling'] == 'strong'
    assert '32768 samples per GPU per step' in d['config']['workload'] and d['config']['global_batch'] == 65536
    assert abs(d['value'] - 65536 / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    assert d['step_mode'].startswith(('eager', 'HIP graph')) and abs(d['config']['mean_ll'] + 1430.6) < 1.0
    assert 'graphed window failed' in d['config']['step_mode'] or d['step_mode'] == 'HIP graph'
