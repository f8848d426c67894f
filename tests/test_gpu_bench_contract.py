"""bench.py's one-line JSON contract on the GPU box: a short run in a child process (the bench spawns its CPU-baseline
processes before it touches HIP, so it has to be its own interpreter), parsed and checked field by field."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '2'] + extra,
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_contract():
    line = _run(['--cpu-seconds', '2'])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
    assert line['n_gpus'] == 1 and line['steps'] == 4 and line['warmup'] == 2 and line['vs_baseline'] is None
    assert isinstance(line['value'], float) and line['value'] > 0
    assert abs(line['value'] - line['config']['restarts_total'] * 4 / (line['ms_per_step'] * 4e-3)) < 1e-6 * line['value']
    r = line['roofline']
    assert r['bound'] in ('hbm', 'mfma', 'fp64_valu') and r['peak'] > 0 and 0 < r['frac'] < 1
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    hb = r.get('hbm_family', r)
    assert hb['bound'] == 'hbm' and 0 < hb['frac'] < 1 and hb['unit'] == 'GB/s'
    c = line['cpu_baseline']
    assert c['kind'] == 'port' and c['value'] > 0 and c['cores'] >= 1 and c['sample']
    assert 'workload' in line['config'] and 'model' not in line['config']


@pytest.mark.gpu
def test_bench_line_without_profile_brackets():
    line = _run(['--no-cpu-baseline', '--no-roofline'])
    assert line['value'] > 0 and line['roofline'] is None and line['cpu_baseline'] is None
