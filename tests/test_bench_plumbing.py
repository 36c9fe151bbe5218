"""bench.py's N > 1 flow on a one-GPU box: `--gpus 2` spawns two ranks itself (torch.distributed.run), both use cuda:0 and exchange
over gloo (VITAE_BENCH_ONE_GPU=1 — a plumbing mode, never a reported number).  Everything else is the real path: rank spawn,
data-parallel step with per-phase graphs and the bucketed exchange, barriers, max-over-ranks clock, rank-0 instrumentation while
the other rank waits, ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_print_one_line():
    env = dict(os.environ, VITAE_BENCH_ONE_GPU='1')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1'],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['rccl_ranks'] == 2 and d['config']['global_batch'] == 8
    assert d['scaling'] == 'weak' and d['value'] > 0 and d['roofline'] is not None and d['cpu_baseline'] is None
    assert 'PLUMBING CHECK' in d['config']['parallelism']


def test_bench_refuses_more_ranks_than_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'VITAE_BENCH_ONE_GPU')}
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1'], capture_output=True,
                       text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and 'refusing' in (r.stderr + r.stdout)
