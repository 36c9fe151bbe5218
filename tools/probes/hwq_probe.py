"""Which HIP streams share a hardware queue?  (MI355X / ROCm 7: GPU_MAX_HW_QUEUES = 4 per process; streams beyond that are
multiplexed, and a long kernel on one of them holds back every other stream on the same queue.)  A spin kernel is put on
stream i and a tiny kernel on stream j: if the tiny one only completes when the spin is over, they share a queue.
Also: where does torch's NCCL (= RCCL) process group run a collective — on its own stream or on the caller's?"""
import os
import time

import torch
import torch.distributed as dist

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()

a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); torch.cuda._sleep(20_000_000); b.record(); torch.cuda.synchronize()
CPS = 20_000_000 / (a.elapsed_time(b) * 1e-3)
SPIN = int(0.03 * CPS)      # 30 ms


def blocked(spin_on, tiny_on, between=None):
    """True if a tiny kernel on ``tiny_on`` waits for a 30 ms spin on ``spin_on``."""
    torch.cuda.synchronize()
    with torch.cuda.stream(spin_on):
        torch.cuda._sleep(SPIN)
    if between is not None:
        between()
    ev = torch.cuda.Event()
    with torch.cuda.stream(tiny_on):
        x.add_(1.0)
        ev.record()
    t0 = time.perf_counter()
    while not ev.query():
        if time.perf_counter() - t0 > 0.2:
            break
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt > 0.010


null = torch.cuda.default_stream(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(10)]
for s in streams:   # first use in creation order
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
names = ['null'] + [f's{i}' for i in range(len(streams))]
alls = [null] + streams
print('rows: spin on; columns: tiny on; X = held back')
print('      ' + ' '.join(f'{n:>4}' for n in names))
for i, si in enumerate(alls):
    row = []
    for j, sj in enumerate(alls):
        row.append('   .' if i == j else ('   X' if blocked(si, sj) else '   -'))
    print(f'{names[i]:>5} ' + ' '.join(row))

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
dist.init_process_group('nccl', device_id=dev)
big = torch.zeros(64 << 20, device=dev)
dist.all_reduce(big)     # communicator + NCCL stream exist now
torch.cuda.synchronize()
for mode in ('async_op=True', 'async_op=False'):
    res = []
    for j, sj in enumerate(alls):
        if sj is streams[0]:
            res.append('   .')
            continue
        works = []

        def coll():
            with torch.cuda.stream(streams[0]):
                w = dist.all_reduce(big, async_op=(mode == 'async_op=True'))
                works.append(w)
        # spin on s0, then a collective issued from s0 (it must wait for the spin): which streams does THAT hold back?
        res.append('   X' if blocked(streams[0], sj, coll) else '   -')
        for w in works:
            if w is not None:
                w.wait()
        torch.cuda.synchronize()
    print(f'collective ({mode}) issued on s0 behind a spin; tiny on: ' + ' '.join(f'{n:>4}' for n in names))
    print('                                                          ' + ' '.join(res))
dist.destroy_process_group()
