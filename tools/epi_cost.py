"""What each epilogue option costs a big-tile forward launch: plain fp32 out / + bias / + residual / + bf16 copy / bf16 only, on a
256x256-tile shape and a 128x128 split-K shape of the batch-32 step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib
from tools.bt_bench import graph_time
dev = 'cuda'
for (M, N, K) in ((3520, 3072, 768), (3520, 768, 3072), (6944, 2048, 512)):
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(4)]
    B = [torch.randn(N, K, device=dev).bfloat16() for _ in range(4)]
    C, C16 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias, res = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
    aux = torch.empty(M, N, device=dev); aux16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    split = lib.vitae_gemm_glds_pick_split_k(M, N, K)
    ws = torch.zeros(1 << 24, device=dev)
    P = lambda t: None if t is None else t.data_ptr()
    cnt = [0]
    def mk(c, c16, b, r, epi=0, ax=None):
        def go():
            cnt[0] += 1; i = cnt[0] % 4
            lib.vitae_gemm_glds(1, 1, P(A[i]), K, P(B[i]), K, P(c), N, P(c16), N, M, N, K, P(b), P(r), N, epi, P(ax), N, 0,
                                1 if epi == 1 else split, P(ws), None, torch.cuda.current_stream().cuda_stream)
        return go
    cases = {'fp32 out': mk(C, None, None, None), '+bias': mk(C, None, bias, None), '+bias+res': mk(C, None, bias, res),
             '+bias+res+bf16': mk(C, C16, bias, res), 'bf16 only +bias': mk(None, C16, bias, None),
             'GELU: aux fp32 + bf16 out': mk(None, C16, bias, None, 1, aux), 'GELU: aux bf16 + bf16 out': mk(None, C16, bias, None, 1 | 16, aux16)}
    print(f'--- M={M} N={N} K={K} split={split} tile={lib.vitae_gemm_glds_bt_choice(1, 1, M, N, K)}')
    for name, go in cases.items():
        go(); torch.cuda.synchronize()
        print(f'  {name:28s} {graph_time(go, 20):7.1f} us', flush=True)
