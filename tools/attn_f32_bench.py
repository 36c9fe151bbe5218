"""fp32 attention (vitae_sdpa_fwd / vitae_sdpa_bwd: the parity mode's and fp32x3's attention) at the bench shapes, graph replay of 20
launches: the fp32 matrix-core kernels (default) against the round-1 VALU kernels (run again with VITAE_ATTN_F32_MFMA=0)."""
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

print('VITAE_ATTN_F32_MFMA =', os.environ.get('VITAE_ATTN_F32_MFMA', '1'))
for B, N, H, hd in ((8, 55, 12, 64), (4, 217, 16, 32), (8, 433, 12, 64), (4, 1729, 16, 32), (64, 55, 12, 64), (32, 217, 16, 32)):
    D = H * hd
    qkv, do = torch.randn(B, N, 3 * D, device='cuda'), torch.randn(B, N, D, device='cuda')
    o, lse = torch.empty(B, N, D, device='cuda'), torch.empty(B, H, N, device='cuda')
    dqkv, delta = torch.empty_like(qkv), torch.empty(B, H, N, device='cuda')
    P = lambda t: t.data_ptr()
    fwd = timeit(lambda st: lib.vitae_sdpa_fwd(P(qkv), P(o), P(lse), B, N, H, hd, st))
    bwd = timeit(lambda st: lib.vitae_sdpa_bwd(P(qkv), P(o), P(do), P(lse), P(dqkv), P(delta), B, N, H, hd, st))
    fl = 4.0 * B * H * N * N * hd
    print(f'B={B} N={N} H={H} hd={hd}: fwd {fwd:.1f} us ({fl / fwd / 1e6:.1f} TFLOP/s)  bwd {bwd:.1f} us ({2.5 * fl / bwd / 1e6:.1f} TFLOP/s)')
