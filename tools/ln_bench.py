"""LayerNorm kernel timings at the bench shapes (graph replay of 20 launches): forward, the atomic backward, the partial-record
backward (VITAE_LN_PART_BLOCKS caps its workgroups) and the reduce launch for 10 such LayerNorms; GB/s on the algorithmic bytes."""
import numpy as np
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
    G = lib.vitae_layernorm_bwd_part_records(M)
    parts = [torch.empty(G * 3 * D, device='cuda') for _ in range(10)]
    u64 = lambda v: np.array(v, dtype=np.uint64)
    a_p, a_w, a_b, a_c = u64([P(t) for t in parts]), u64([P(dw)] * 10), u64([P(db)] * 10), u64([P(cs)] * 10)
    a_g, a_d = np.array([G] * 10, dtype=np.int32), np.array([D] * 10, dtype=np.int32)
    tp = timeit(lambda st: lib.vitae_layernorm_bwd_part(P(dy), P(x), P(w), P(mean), P(rstd), P(dx), P(parts[0]), P(dx16), M, D, 1, st))
    tr = timeit(lambda st: lib.vitae_ln_grad_reduce(10, a_p.ctypes.data, a_w.ctypes.data, a_b.ctypes.data, a_c.ctypes.data, a_g.ctypes.data, a_d.ctypes.data, st))
    bwd_bytes = M * D * (4 * 4 + 2)
    print(M, D, 'records', G, 'bwd-part %.1f (%.0f GB/s)' % (tp, bwd_bytes / tp / 1e3), 'reduce x10 %.1f' % tr,
          'fwd %.1f' % timeit(lambda st: lib.vitae_layernorm_fwd(P(x), P(w), P(b), None, P(y16), P(mean), P(rstd), M, D, 1e-6, st)),
          'bwd %.1f' % timeit(lambda st: lib.vitae_layernorm_bwd(P(dy), P(x), P(w), P(mean), P(rstd), P(dx), P(dw), P(db), P(dx16), P(cs), M, D, 1, st)),
          'colsum[M,3D] %.1f' % timeit(lambda st: lib.vitae_colsum_accum(P(big), 3 * D, P(csb), M, 3 * D, st)),
          'zero %.1f' % timeit(lambda st: lib.vitae_memset_zero(P(cs), D * 4, st)))
