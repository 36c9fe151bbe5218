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
    # the dual-leg flow: the host-issued exchange was timed and described; the native (RCCL in the graph) leg cannot run on a shared GPU
    assert d['config']['exchange']['route'].startswith('host-issued') and d['config']['exchange']['buckets'] >= 2
    assert 'skipped' in d['config']['also_exchange']


def test_bench_refuses_more_ranks_than_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'VITAE_BENCH_ONE_GPU')}
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1'], capture_output=True,
                       text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and 'refusing' in (r.stderr + r.stdout)


def test_bench_times_both_exchange_routes_on_a_world_of_one():
    """at N > 1 bench.py times BOTH gradient-exchange routes in one invocation — torch.distributed collectives
    between per-phase graphs, and RCCL through the C ABI inside the one step graph — reports the faster as `value` and the other
    under config.also_exchange.  On a one-GPU box the same flow runs over a communicator of one rank (VITAE_FORCE_DDP=1)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'VITAE_BENCH_ONE_GPU', 'RANK', 'LOCAL_RANK')}
    env['VITAE_FORCE_DDP'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--profile-steps', '0'],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    a, b = d['config']['exchange'], d['config']['also_exchange']
    assert {a['route'].split(':')[0], b['route'].split(':')[0]} == {'host-issued', 'native'}, (a, b)
    nat = a if a['route'].startswith('native') else b
    assert nat['vitae_ddp_world_size'] == 1 and d['config']['rccl_ranks'] == 1
    assert a['ms_per_step'] <= b['ms_per_step'] and abs(d['ms_per_step'] - a['ms_per_step']) < 1e-3
    assert a['wire_dtype'] == 'bf16' and len(a['bucket_mbytes']) == a['buckets']
    # the diagnostics of the one hardware run: per-bucket all-reduce time (host-issued route), the step with the exchange off
    host = a if a['route'].startswith('host-issued') else b
    dg = host['diagnostics']
    assert 'error' not in dg, dg
    assert len(dg['allreduce_ms_per_bucket']) == host['buckets'] and dg['ms_per_step_exchange_off'] > 0
    assert 'exposed_exchange_ms_per_step' in dg and 'ms_per_step_exchange_off' in nat['diagnostics']
