"""The driver's contract on bench.py: one JSON line on stdout with the agreed keys, a roofline and a cpu_baseline block,
internally consistent numbers.  A short run of the real script on the GPU (secondary configurations skipped)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *flags], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
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
    assert r['kernel_ms'] <= d['ms_per_step'] * 1.05          # the kernel cannot take longer than the step around it
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and d['value'] > 10 * c['value']
    # the mean LL of N(0,1) inputs under this model: finite and the same from run to run (seeded)
    assert abs(d['config']['mean_ll'] + 1430.6) < 1.0
