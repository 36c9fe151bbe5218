"""Attention kernel timings at the bench shapes (graph replay of 20 launches), bf16 q | k | v input (the step's path): forward,
backward (one launch when the head fits LDS, else the dQ + dK/dV streaming kernels), with TFLOP/s on 4 N^2 hd (forward) and
10 N^2 hd (backward) per head.  VITAE_ATTN_RB=1|2 forces the 32-row blocks per wave of the streaming kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd import _abi
lib = _abi.lib

def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn(s.cuda_stream)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn(s.cuda_stream)
        g.replay(); s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); g.replay(); b.record(s); s.synchronize()
    return a.elapsed_time(b) / (2 * n) * 1e3

shapes = ((8, 55, 12, 64), (4, 217, 16, 32), (8, 433, 12, 64), (4, 1729, 16, 32), (64, 55, 12, 64), (32, 217, 16, 32))
for B, N, H, hd in shapes:
    D = H * hd
    qkv, do = torch.randn(B, N, 3 * D, device='cuda'), torch.randn(B, N, D, device='cuda')
    q16 = qkv.bfloat16()
    o, lse = torch.empty(B, N, D, device='cuda'), torch.empty(B, H, N, device='cuda')
    o16 = torch.empty(B, N, D, dtype=torch.bfloat16, device='cuda')
    d16 = torch.empty(B, N, 3 * D, dtype=torch.bfloat16, device='cuda')
    cs, delta = torch.zeros(3 * D, device='cuda'), torch.empty(B, H, N, device='cuda')
    P = lambda t: t.data_ptr()
    fwd = timeit(lambda st: lib.vitae_sdpa_mfma_fwd_bf16in(P(q16), P(o), P(o16), P(lse), B, N, H, hd, st))
    bwd = timeit(lambda st: lib.vitae_sdpa_mfma_bwd_bf16in(P(q16), P(o), P(do), P(lse), None, P(d16), None, P(delta), B, N, H, hd, st))
    ff, fb = 4.0 * B * H * N * N * hd, 10.0 * B * H * N * N * hd
    print(f'B={B} N={N} H={H} hd={hd}: fwd {fwd:.1f} us ({ff / fwd / 1e6:.0f} TF/s)  bwd {bwd:.1f} us ({fb / bwd / 1e6:.0f} TF/s)  '
          f'fwd+bwd {(ff + fb) / (fwd + bwd) / 1e6:.0f} TF/s', flush=True)
