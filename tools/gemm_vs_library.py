"""Calibration, not product: what does the vendor library (torch.matmul -> hipBLASLt / rocBLAS) need for the forward GEMM shapes
of the step, next to csrc/gemm_glds.hip's launcher on the same operands?  Both timed as 200 back-to-back launches between two
events on one stream (so each figure contains the ~2 us launch floor); several operand sets are cycled so L2 is not pre-warmed
with exactly the same weights each time.  y[M,N] = x[M,K] @ W[N,K]^T, bf16 in, fp32 out."""
import sys
import torch
from vit_ae_plus_plus_amd._abi import lib

dev = torch.device('cuda', 0)
st = torch.cuda.current_stream(dev).cuda_stream
SHAPES = []
for B in (4, 8, 32):
    Me, Md = B * 2 * 55, B * 217
    SHAPES += [(f'B{B} enc qkv', Me, 2304, 768), (f'B{B} enc proj', Me, 768, 768), (f'B{B} enc fc1', Me, 3072, 768),
               (f'B{B} enc fc2', Me, 768, 3072), (f'B{B} dec qkv', Md, 1536, 512), (f'B{B} dec fc1', Md, 2048, 512),
               (f'B{B} dec fc2', Md, 512, 2048), (f'B{B} dec pred', Md, 16384, 512), (f'B{B} patch embed', B * 2 * 54, 768, 16384)]
NSET, REP = 4, 200
ws = torch.zeros(64 << 20, device=dev)
print(f'{"shape":<18}{"M":>6}{"N":>6}{"K":>6} | {"library us":>10} {"TF/s":>7} | {"glds us":>8} {"TF/s":>7} | ratio')
for name, M, N, K in SHAPES:
    xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NSET)]
    wsets = [torch.randn(N, K, device=dev).bfloat16() for _ in range(NSET)]
    y = torch.empty(M, N, device=dev)
    y16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def lib_call(i):
        torch.matmul(xs[i % NSET], wsets[i % NSET].t(), out=y16)

    s = lib.vitae_gemm_glds_pick_split_k(M, N, K)

    def glds_call(i):
        lib.vitae_gemm_glds(1, 1, xs[i % NSET].data_ptr(), K, wsets[i % NSET].data_ptr(), K, y.data_ptr(), N, None, 0, M, N, K, None,
                            None, 0, 0, None, 0, 0, s, ws.data_ptr(), None, st)

    def glds16_call(i):      # like for like with the library: bf16 output only (what the step's qkv / fc1 launches write)
        lib.vitae_gemm_glds(1, 1, xs[i % NSET].data_ptr(), K, wsets[i % NSET].data_ptr(), K, None, N, y16.data_ptr(), N, M, N, K, None,
                            None, 0, 0, None, 0, 0, s, ws.data_ptr(), None, st)

    out = []
    for f in (lib_call, glds_call, glds16_call):
        for i in range(10):
            f(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(REP):
            f(i)
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / REP * 1e3)
    fl = 2.0 * M * N * K
    print(f'{name:<18}{M:>6}{N:>6}{K:>6} | {out[0]:>10.1f} {fl / out[0] / 1e6:>7.0f} | {out[1]:>8.1f} {fl / out[1] / 1e6:>7.0f} | {out[1] / out[0]:.2f} | bf16 out {out[2]:>8.1f} {out[2] / out[0]:.2f}')
