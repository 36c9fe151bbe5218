"""Micro-benchmark of vitae_gemm_bf16 on the GEMM shapes of BASELINE config 2 (B=4, contrastive)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib

def graph_time(go, iters):
    """GPU time per launch with the launches replayed from a HIP graph (eager python launches are host bound)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(iters): go_on_current(go)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3


def go_on_current(go):
    go()


COLD = '--cold' in sys.argv


def ncopies(nbytes):
    return max(1, min(64, int(400e6 // max(nbytes, 1)))) if COLD else 1


def run(name, form, M, N, K, b16, iters=50):
    """form: 'fwd' (A[M,K] kc, B[N,K] kc), 'dgrad' (A[M,K] kc, B stored [K,N] row-contig), 'wgrad' (A stored [K,M], B stored [K,N])."""
    dev = 'cuda'
    akc, bkc = {'fwd': (1, 1), 'dgrad': (1, 0), 'wgrad': (0, 0)}[form]
    A = torch.randn((M, K) if akc else (K, M), device=dev)
    Bt = torch.randn((N, K) if bkc else (K, N), device=dev)
    Bp = Bt.to(torch.bfloat16) if b16 else Bt
    Bs = [Bp] + [Bp.clone() for _ in range(ncopies(Bp.numel() * Bp.element_size()) - 1)]
    cnt = [0]
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(1 << 24, device=dev)
    lda = K if akc else M
    ldb = K if bkc else N
    split = lib.vitae_gemm_bf16_pick_split_k(M, N, K)
    def go():
        cnt[0] += 1
        Bp = Bs[cnt[0] % len(Bs)]
        lib.vitae_gemm_bf16(akc, bkc, A.data_ptr(), lda, Bp.data_ptr(), ldb, b16, C.data_ptr(), N, M, N, K, None, None, 0, 0, None, 0, 0,
                            split, ws.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    for _ in range(5): go()
    torch.cuda.synchronize()
    us = graph_time(go, iters)
    print(f'{name:28s} {form:5s} M={M:5d} N={N:5d} K={K:5d} b16={b16} split={split:2d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s')
    return us

def run_glds(name, form, M, N, K, iters=50, check=True):
    dev = 'cuda'
    akc, bkc = {'fwd': (1, 1), 'dgrad': (1, 0), 'wgrad': (0, 0)}[form]
    Kp = (K + 63) // 64 * 64
    A = torch.zeros((M, Kp) if akc else (Kp, M), device=dev)
    Bt = torch.zeros((N, Kp) if bkc else (Kp, N), device=dev)
    if akc: A[:, :K] = torch.randn(M, K, device=dev)
    else: A[:K] = torch.randn(K, M, device=dev)
    if bkc: Bt[:, :K] = torch.randn(N, K, device=dev)
    else: Bt[:K] = torch.randn(K, N, device=dev)
    A16, B16_0 = A.to(torch.bfloat16), Bt.to(torch.bfloat16)
    B16 = B16_0
    Bs = [B16_0] + [B16_0.clone() for _ in range(ncopies(B16_0.numel() * 2) - 1)]
    cnt = [0]
    C = torch.full((M, N), float('nan'), device=dev)
    ws = torch.zeros(1 << 24, device=dev)
    lda = Kp if akc else M
    ldb = Kp if bkc else N
    split = lib.vitae_gemm_glds_pick_split_k(M, N, Kp)
    RES = 'resid' in sys.argv
    bias = torch.randn(N, device=dev)
    As = [A16] + [A16.clone() for _ in range(len(Bs) - 1)] if RES else [A16]
    Rs = [torch.randn(M, N, device=dev) for _ in range(len(Bs))] if RES else [None]
    Cs = [torch.empty(M, N, device=dev) for _ in range(len(Bs))] if RES else [C]
    def go():
        cnt[0] += 1
        B16 = Bs[cnt[0] % len(Bs)]
        i = cnt[0] % len(As)
        lib.vitae_gemm_glds(akc, bkc, As[i].data_ptr(), lda, B16.data_ptr(), ldb, Cs[i % len(Cs)].data_ptr(), N, None, 0, M, N, Kp,
                            bias.data_ptr() if RES else None, Rs[i % len(Rs)].data_ptr() if RES else None, N, 0,
                            None, 0, 0, split, ws.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    for _ in range(5): go()
    torch.cuda.synchronize()
    err = ''
    if check:
        Af = A16.float() if akc else A16.float().t()
        Bf = B16.float() if bkc else B16.float().t()
        ref = Af @ Bf.t()
        e = float((C - ref).abs().max() / (ref.abs().max() + 1e-20))
        err = f' relerr {e:.1e}' + (' !!!' if not (e < 2e-3) else '')
    us = graph_time(go, iters)
    print(f'GLDS {name:24s} {form:5s} M={M:5d} N={N:5d} K={K:5d} split={split:2d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s{err}')
    return us


def run_pair(name, M, N, K, iters=50):
    """Backward of Linear(K -> N) on M tokens: dx[M,K] = dy16 @ W16, dW[N,K] = dy16^T @ x16 in one launch."""
    dev = 'cuda'
    Mp = (M + 63) // 64 * 64
    dy16 = torch.zeros(Mp, N, dtype=torch.bfloat16, device=dev); dy16[:M] = torch.randn(M, N, device=dev)
    x16 = torch.zeros(Mp, K, dtype=torch.bfloat16, device=dev); x16[:M] = torch.randn(M, K, device=dev)
    w0 = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    Ws = [w0] + [w0.clone() for _ in range(ncopies(w0.numel() * 2) - 1)]
    dx, dw = torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
    cnt = [0]
    split = lib.vitae_linear_bwd_pair_pick_split_k(M, Mp, N, K)
    ws = torch.zeros(1 << 23, device=dev)
    def go():
        cnt[0] += 1
        w = Ws[cnt[0] % len(Ws)]
        lib.vitae_linear_bwd_pair_glds(dy16.data_ptr(), w.data_ptr(), x16.data_ptr(), dx.data_ptr(), None, dw.data_ptr(), None, M, Mp, N, K,
                                       0, None, None, None, 0, 0, split, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    for _ in range(5): go()
    torch.cuda.synchronize()
    e1 = float((dx - dy16[:M].float() @ w0.float()).abs().max() / (dx.abs().max() + 1e-20))
    e2 = float((dw - dy16.float().t() @ x16.float()).abs().max() / (dw.abs().max() + 1e-20))
    us = graph_time(go, iters)
    print(f'PAIR {name:24s} M={M:5d} N={N:5d} K={K:5d} split={split}  {us:8.1f} us  {4.0*M*N*K/us/1e6:7.1f} TF/s  err {e1:.1e} {e2:.1e}' + (' !!!' if max(e1, e2) > 2e-3 else ''))
    return us


if __name__ == '__main__':
    if 'pair' in sys.argv:
        tot = 0.0
        for (nm, M, d, h, cnt) in (('enc', 440, 768, 3072, 12), ('dec', 868, 512, 2048, 8)):
            for (n2, N, K) in (('qkv', 3 * d, d), ('proj', d, d), ('fc1', h, d), ('fc2', d, h)):
                tot += cnt * run_pair(f'{nm}.{n2}', M, N, K)
        tot += run_pair('pred', 868, 16384, 512)
        print(f'weighted total {tot/1e3:.3f} ms per step')
        sys.exit(0)
    glds = 'glds' in sys.argv
    tot = 0.0
    E, D_ = 440, 868
    shapes = []
    for (nm, M, d, h, cnt) in (('enc', 440, 768, 3072, 12), ('dec', 868, 512, 2048, 8)):
        for (n2, N, K) in (('qkv', 3 * d, d), ('proj', d, d), ('fc1', h, d), ('fc2', d, h)):
            shapes.append((f'{nm}.{n2}', 'fwd', M, N, K, 1, cnt))
            shapes.append((f'{nm}.{n2}', 'dgrad', M, K, N, 1, cnt))
            shapes.append((f'{nm}.{n2}', 'wgrad', N, K, M, 0, cnt))
    shapes += [('patch_embed', 'fwd', 432, 768, 16384, 1, 1), ('patch_embed', 'wgrad', 768, 16384, 432, 0, 1),
               ('pred', 'fwd', 868, 16384, 512, 1, 1), ('pred', 'dgrad', 868, 512, 16384, 1, 1), ('pred', 'wgrad', 16384, 512, 868, 0, 1),
               ('dec_embed', 'fwd', 220, 512, 768, 1, 1), ('predictor', 'fwd', 440, 768, 768, 1, 2)]
    for nm, form, M, N, K, b16, cnt in shapes:
        tot += cnt * (run_glds(nm, form, M, N, K) if glds else run(nm, form, M, N, K, b16))
    print(f'weighted total {tot/1e3:.3f} ms per step')
