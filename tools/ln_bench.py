"""LayerNorm / colsum kernel timings at the bench shapes (graph replay of 20 launches)."""
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

for M, D in ((440, 768), (868, 512), (880, 768), (1736, 512), (3520, 768), (6944, 512), (3464, 768), (6916, 512)):
    x, dy = torch.randn(M, D, device='cuda'), torch.randn(M, D, device='cuda')
    w, b = torch.randn(D, device='cuda'), torch.randn(D, device='cuda')
    y, y16 = torch.empty(M, D, device='cuda'), torch.empty(M, D, dtype=torch.bfloat16, device='cuda')
    mean, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
    dx, dx16 = torch.zeros(M, D, device='cuda'), torch.empty(M, D, dtype=torch.bfloat16, device='cuda')
    dw, db, cs = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
    big = torch.randn(M, 3 * D, device='cuda'); csb = torch.zeros(3 * D, device='cuda')
    P = lambda t: t.data_ptr()
    print(M, D,
          'fwd %.1f' % timeit(lambda st: lib.vitae_layernorm_fwd(P(x), P(w), P(b), None, P(y16), P(mean), P(rstd), M, D, 1e-6, st)),
          'bwd %.1f' % timeit(lambda st: lib.vitae_layernorm_bwd(P(dy), P(x), P(w), P(mean), P(rstd), P(dx), P(dw), P(db), P(dx16), P(cs), M, D, 1, st)),
          'colsum[M,3D] %.1f' % timeit(lambda st: lib.vitae_colsum_accum(P(big), 3 * D, P(csb), M, 3 * D, st)),
          'zero %.1f' % timeit(lambda st: lib.vitae_memset_zero(P(cs), D * 4, st)))
