"""vitae_gemm_wsx3 against the older fp32x3 kernel (vitae_gemm at VITAE_PREC_BF16X3 + its split-K reduce launch) on batch-4 shapes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from vit_ae_plus_plus_amd._abi import lib
from bt_bench import graph_time
dev = 'cuda'
st = lambda: torch.cuda.current_stream().cuda_stream
for form, M, N, K in (('fwd', 440, 2304, 768), ('fwd', 440, 768, 768), ('fwd', 440, 768, 3072), ('dgrad', 440, 768, 2304), ('dgrad', 440, 3072, 768),
                      ('wgrad', 768, 768, 440), ('wgrad', 3072, 768, 440), ('wgrad', 2048, 512, 868), ('fwd', 868, 2048, 512)):
    akc, bkc = {'fwd': (1, 1), 'dgrad': (1, 0), 'wgrad': (0, 0)}[form]
    NS = 4
    As = [torch.randn((M, K) if akc else (K, M), device=dev) for _ in range(NS)]
    Bs = [torch.randn((N, K) if bkc else (K, N), device=dev) for _ in range(NS)]
    C = torch.empty(M, N, device=dev)
    ws = torch.zeros(1 << 23, device=dev); ws2 = torch.empty(1 << 23, device=dev)
    lda, ldb = (K if akc else M), (K if bkc else N)
    cnt = [0]
    for s_force in (None, 1):
        s = lib.vitae_gemm_wsx3_pick_split_k(M, N, K) if s_force is None else s_force
        def new():
            cnt[0] += 1; i = cnt[0] % NS
            lib.vitae_gemm_wsx3(akc, bkc, As[i].data_ptr(), lda, Bs[i].data_ptr(), ldb, C.data_ptr(), N, M, N, K, None, None, 0, 0, None, 0, 0, s, ws.data_ptr(), None, None, st())
        new(); torch.cuda.synchronize()
        print(f'{form:5s} {M:5d} {N:5d} {K:5d} wsx3 split {s}: {graph_time(new, 20):6.1f} us', flush=True)
    so = lib.vitae_gemm_bf16x3_pick_split_k(M, N, K)
    def old():
        cnt[0] += 1; i = cnt[0] % NS
        lib.vitae_gemm(2, akc, bkc, As[i].data_ptr(), lda, Bs[i].data_ptr(), ldb, C.data_ptr(), N, M, N, K, None, None, 0, 0, None, 0, 0, so, ws2.data_ptr(), st())
    old(); torch.cuda.synchronize()
    print(f'{form:5s} {M:5d} {N:5d} {K:5d} old  split {so}: {graph_time(old, 20):6.1f} us', flush=True)
