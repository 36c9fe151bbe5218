"""A few eager launches of one Linear-backward pair (for rocprofv3 --pmc, which must not see graph replays):
python tools/pair_once.py M N K [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd._abi import lib

M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
Mp = (M + 63) // 64 * 64
dev = 'cuda'
dy16 = torch.zeros(Mp, N, dtype=torch.bfloat16, device=dev); dy16[:M] = torch.randn(M, N, device=dev)
x16 = torch.zeros(Mp, K, dtype=torch.bfloat16, device=dev); x16[:M] = torch.randn(M, K, device=dev)
w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
dx, dw = torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
ws = torch.zeros(1 << 23, device=dev)
split = lib.vitae_linear_bwd_pair_pick_split_k(M, Mp, N, K)
for _ in range(reps):
    lib.vitae_linear_bwd_pair_glds(dy16.data_ptr(), w.data_ptr(), x16.data_ptr(), dx.data_ptr(), None, dw.data_ptr(), None, M, Mp, N, K,
                                   0, None, None, None, 0, 0, split, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print('split', split)
