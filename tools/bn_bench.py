"""Predictor BatchNorm1d + ReLU (csrc/norm.hip): one 64-column strip per workgroup against the row-split form (two launches), forward and
backward, at the rows-per-view of batch 4 / 8 / 32 and patch 8 (D = 768): python tools/bn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib
from bt_bench import graph_time
D = 768
for R in (220, 440, 880, 1732, 1760):
    x, dy = torch.randn(R, D, device='cuda'), torch.randn(R, D, device='cuda')
    w, b = torch.rand(D, device='cuda') + 0.5, torch.randn(D, device='cuda')
    y, dx = torch.empty_like(x), torch.empty_like(x)
    y16, dx16 = torch.empty(R, D, dtype=torch.bfloat16, device='cuda'), torch.empty(R, D, dtype=torch.bfloat16, device='cuda')
    sm, sr, rm, rv = (torch.zeros(D, device='cuda') for _ in range(4))
    dw, db = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
    nbt = torch.zeros((), dtype=torch.int64, device='cuda')
    ws = torch.empty(int(lib.vitae_bn1d_split_ws_floats(R, D)), device='cuda')
    P = lambda t: t.data_ptr()
    st = lambda: torch.cuda.current_stream().cuda_stream
    fa = (P(x), P(w), P(b), P(y), P(y16), P(sm), P(sr), P(rm), P(rv), P(nbt), R, D, 1e-5, 0.1)
    ba = (P(dy), P(x), P(y), P(w), P(sm), P(sr), P(dx), P(dx16), P(dw), P(db), R, D)
    t = {'fwd strip': graph_time(lambda: lib.vitae_bn1d_relu_fwd(*fa, st()), 20),
         'fwd split': graph_time(lambda: lib.vitae_bn1d_relu_fwd_split(*fa, P(ws), st()), 20),
         'bwd strip': graph_time(lambda: lib.vitae_bn1d_relu_bwd(*ba, st()), 20),
         'bwd split': graph_time(lambda: lib.vitae_bn1d_relu_bwd_split(*ba, P(ws), st()), 20)}
    print(f'R={R:5d} D={D}: ' + '  '.join(f'{k} {v:6.1f} us' for k, v in t.items()), flush=True)
