"""Big-tile GEMM family (csrc/gemm_bt.hip): correctness against an fp32 product of the same bf16 operands and GPU time per launch,
every tile x every operand form, on the shapes of the step at batch 4 / 8 / 32 and on square calibration shapes.
    python tools/bt_bench.py [big] [step] [forms=fwd,dgrad,wgrad] [tiles=0,1,2,3,-2]
Tile -2 = the 64-row family (the previous default), -1 = the cost model's pick."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib

dev = 'cuda'
NAMES = {0: '256x256', 1: '256x128', 2: '128x256', 3: '128x128', 4: 'ws128', 5: 'ws64', 6: 'ws128x256', -2: '64-row', -1: 'auto'}


def graph_time(go, iters):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(iters):
            go()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def one(form, M, N, K, tile, iters=20, check=True, epi=False):
    akc, bkc = {'fwd': (1, 1), 'dgrad': (1, 0), 'wgrad': (0, 0)}[form]
    g = torch.Generator(device=dev).manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((M, K) if akc else (K, M), device=dev, generator=g).bfloat16()
    B = torch.randn((N, K) if bkc else (K, N), device=dev, generator=g).bfloat16()
    NS = max(1, min(8, int(300e6 // max(1, (A.numel() + B.numel()) * 2))))
    As = [A] + [A.clone() for _ in range(NS - 1)]
    Bs = [B] + [B.clone() for _ in range(NS - 1)]
    C = torch.full((M, N), float('nan'), device=dev)
    bias = torch.randn(N, device=dev, generator=g) if epi else None
    res = torch.randn(M, N, device=dev, generator=g) if epi else None
    C16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi else None
    lib.vitae_gemm_glds_set_bt_tile(tile)
    got = lib.vitae_gemm_glds_bt_choice(akc, bkc, M, N, K)
    split = lib.vitae_gemm_glds_pick_split_k_form(akc, bkc, M, N, K)
    ws = torch.zeros(1 << 24, device=dev)
    cnt = [0]

    def go():
        cnt[0] += 1
        i = cnt[0] % NS
        lib.vitae_gemm_glds(akc, bkc, As[i].data_ptr(), K if akc else M, Bs[i].data_ptr(), K if bkc else N, C.data_ptr(), N,
                            C16.data_ptr() if epi else None, N, M, N, K, bias.data_ptr() if epi else None,
                            res.data_ptr() if epi else None, N, 0, None, 0, 0, split, ws.data_ptr(), None,
                            torch.cuda.current_stream().cuda_stream)
    go(); torch.cuda.synchronize()
    err = ''
    if check:
        Af = A.float() if akc else A.float().t()
        Bf = B.float() if bkc else B.float().t()
        ref = Af @ Bf.t()
        if epi:
            ref = ref + bias + res
        e = float((C - ref).abs().max() / (ref.abs().max() + 1e-20))
        err = f' relerr {e:.1e}' + (' !!!' if not (e < 2e-3) else '')
        if epi:
            e16 = float((C16.float() - ref).abs().max() / (ref.abs().max() + 1e-20))
            err += f' c16 {e16:.1e}' + (' !!!' if not (e16 < 1e-2) else '')
    us = graph_time(go, iters)
    lib.vitae_gemm_glds_set_bt_tile(-1)
    print(f'{form:5s} M={M:5d} N={N:5d} K={K:5d} tile {NAMES[tile]:>9s}->{got:2d} split={split:2d} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s{err}', flush=True)
    return us


if __name__ == '__main__':
    args = sys.argv[1:]
    forms = next((a.split('=')[1].split(',') for a in args if a.startswith('forms=')), ['fwd'])
    tiles = next(([int(x) for x in a.split('=')[1].split(',')] for a in args if a.startswith('tiles=')), [0, 3, -2, -1])
    if 'big' in args:
        for form in forms:
            for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 4096), (2048, 2048, 2048)):
                for t in tiles:
                    one(form, M, N, K, t, iters=5, check=(M <= 4096))
    if 'ragged' in args:
        for form in forms:
            for (M, N, K) in ((868, 16384, 512), (440, 772, 192), (1000, 520, 128), (300, 264, 640)):
                for t in tiles:
                    one(form, M, N, K, t, iters=5, epi=True)
    if 'step' in args:
        for Bt in (4, 8, 32):
            Me, Md = Bt * 2 * 55, Bt * 217
            shapes = [('enc qkv', Me, 2304, 768), ('enc proj', Me, 768, 768), ('enc fc1', Me, 3072, 768), ('enc fc2', Me, 768, 3072),
                      ('dec qkv', Md, 1536, 512), ('dec fc1', Md, 2048, 512), ('dec fc2', Md, 512, 2048), ('dec pred', Md, 16384, 512),
                      ('patch embed', Bt * 2 * 54, 768, 16384)]
            for name, M, N, K in shapes:
                print(f'--- B{Bt} {name}')
                for form in forms:
                    m, n, k = (M, N, K) if form == 'fwd' else (M, K, N) if form == 'dgrad' else (N, K, (M + 63) // 64 * 64)
                    for t in tiles:
                        one(form, m, n, k, t, iters=20)
