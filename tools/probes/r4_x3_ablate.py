import sys, os
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import torch
from vit_ae_plus_plus_amd._abi import lib
from bt_bench import graph_time
st = lambda: torch.cuda.current_stream().cuda_stream
for K in (768, 3072):
    M, N = 440, 768
    As = [torch.randn(M, K, device='cuda') for _ in range(4)]; Bs = [torch.randn(N, K, device='cuda') for _ in range(4)]
    C = torch.empty(M, N, device='cuda'); ws = torch.zeros(1 << 22, device='cuda'); cnt = [0]
    def new():
        cnt[0] += 1; i = cnt[0] % 4
        lib.vitae_gemm_wsx3(1, 1, As[i].data_ptr(), K, Bs[i].data_ptr(), K, C.data_ptr(), N, M, N, K, None, None, 0, 0, None, 0, 0, 1, ws.data_ptr(), None, None, st())
    new(); torch.cuda.synchronize()
    print(f'K={K}: {graph_time(new, 20):6.1f} us', flush=True)
