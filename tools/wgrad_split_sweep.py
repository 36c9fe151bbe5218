"""Weight-gradient form (both operands row-contiguous) on the 64-row tiles with an explicit k-split: GPU time per launch on the
shapes of the batch-32 and patch-8 steps.  python tools/wgrad_split_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib
from bt_bench import graph_time

dev = 'cuda'
SHAPES = [('B32 dec qkv', 1536, 512, 6976), ('B32 dec proj', 512, 512, 6976), ('B32 dec fc1', 2048, 512, 6976), ('B32 dec fc2', 512, 2048, 6976),
          ('B32 enc qkv', 2304, 768, 3520), ('B32 enc proj', 768, 768, 3520), ('B32 enc fc1', 3072, 768, 3520), ('B32 enc fc2', 768, 3072, 3520),
          ('p8 dec qkv', 1536, 512, 6976), ('p8 enc qkv', 2304, 768, 3520), ('B8 enc fc1', 3072, 768, 896), ('B8 dec fc1', 2048, 512, 1792)]
ws = torch.zeros(1 << 24, device=dev)
for name, M, N, K in SHAPES:
    A = torch.randn(K, M, device=dev).bfloat16()
    B = torch.randn(K, N, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev)
    row = f'{name:14s} M={M:5d} N={N:5d} K={K:5d} |'
    for tile, splits in ((-2, (1, 2, 3, 4, 6)), (3, (0,)), (-1, (0,))):
        lib.vitae_gemm_glds_set_bt_tile(tile)
        for s in splits:
            sp = s if s else lib.vitae_gemm_glds_pick_split_k(M, N, K)
            go = lambda: lib.vitae_gemm_glds(0, 0, A.data_ptr(), M, B.data_ptr(), N, C.data_ptr(), N, None, 0, M, N, K, None, None, 0, 0, None, 0,
                                             0, sp, ws.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            go(); torch.cuda.synchronize()
            us = graph_time(go, 20)
            row += f' {"64r" if tile == -2 else ("bt128" if tile == 3 else "auto")}/s{sp}: {us:5.1f}'
    print(row, flush=True)
lib.vitae_gemm_glds_set_bt_tile(-1)
