"""bench.py prints ONE JSON line with the contract fields (task statement) plus `roofline` and `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_line_contract(dev):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--designs', '2',
           '--nodes', '4096', '--levels', '16', '--tile', '64', '--batch-paths', '64']
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f'expected one stdout line, got {len(lines)}'
    j = json.loads(lines[0])
    for key, typ in (('metric', str), ('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int),
                     ('ms_per_step', float), ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str),
                     ('config', dict), ('roofline', dict), ('cpu_baseline', dict)):
        assert isinstance(j[key], typ), key
    assert 'vs_baseline' in j and j['vs_baseline'] is None          # BASELINE.md holds no published number
    assert j['n_gpus'] == 1 and j['steps'] == 3 and j['warmup'] == 1 and j['scaling'] == 'weak' and j['dtype'] == 'bf16'
    f32 = j['heldout_eval_f32_same_schedule']                        # the same schedule replayed in exact fp32
    assert f32 is not None and f32['ms_per_step'] > 0 and abs(f32['mae_drift']) < 1.0
    assert j['higher_is_better'] is True and 'workload' in j['config'] and j['value'] > 0
    assert abs(j['value'] - 2 * 3 / (j['ms_per_step'] * 3 / 1e3)) / j['value'] < 1e-6      # designs / s over the timed steps
    r = j['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and r['peak'] > 0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 'traffic' in r
    c = j['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and isinstance(c['sample'], str)
