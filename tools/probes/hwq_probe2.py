"""On which stream does torch's NCCL (= RCCL) process group run a collective?  World of one: an out-of-place all-gather is a
device copy of the whole buffer (4 GB -> a few ms) on the stream RCCL was given; tiny kernels on candidate streams that only
finish when the copy is over share its hardware queue."""
import os
import time

import torch
import torch.distributed as dist

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
x = torch.zeros(1024, device=dev)
null = torch.cuda.default_stream(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(10)]
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
names = ['null'] + [f's{i}' for i in range(len(streams))]
alls = [null] + streams
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29578')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
dist.init_process_group('nccl', device_id=dev)
src = torch.zeros(1 << 30, device=dev)
dst = torch.empty(1 << 30, device=dev)
dist.all_gather_into_tensor(dst, src)
torch.cuda.synchronize()


def shot(issue_on, async_op):
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in alls]
    t0 = torch.cuda.Event(enable_timing=True)
    done = torch.cuda.Event(enable_timing=True)
    t0.record(issue_on)
    for s in alls:
        s.wait_event(t0)
    with torch.cuda.stream(issue_on):
        w = dist.all_gather_into_tensor(dst, src, async_op=async_op)
    for s, (a, b) in zip(alls, evs):
        if s is issue_on:
            continue
        with torch.cuda.stream(s):
            x.add_(1.0)
            b.record()
    with torch.cuda.stream(issue_on):
        if w is not None:
            w.wait()
        done.record()
    torch.cuda.synchronize()
    tot = t0.elapsed_time(done)
    lat = ['   . ' if s is issue_on else f'{t0.elapsed_time(b):5.2f}' for s, (a, b) in zip(alls, evs)]
    return tot, lat


for async_op in (True, False):
    for issue in (streams[0], streams[1], null):
        tot, lat = shot(issue, async_op)
        print(f'async_op={async_op} issued on {names[alls.index(issue)]:>4}: copy done after {tot:5.2f} ms; tiny-kernel completion (ms) on')
        print('     ' + ' '.join(f'{n:>5}' for n in names))
        print('     ' + ' '.join(lat))
dist.destroy_process_group()
