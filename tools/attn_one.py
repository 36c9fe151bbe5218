"""A few launches of the attention kernels at one shape (for rocprofv3 --pmc passes): python tools/attn_one.py B N H hd [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd import _abi
lib = _abi.lib
B, N, H, hd = (int(x) for x in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
D = H * hd
q16 = torch.randn(B, N, 3 * D, device='cuda').bfloat16()
do = torch.randn(B, N, D, device='cuda')
o, lse = torch.empty(B, N, D, device='cuda'), torch.empty(B, H, N, device='cuda')
o16 = torch.empty(B, N, D, dtype=torch.bfloat16, device='cuda')
d16 = torch.empty(B, N, 3 * D, dtype=torch.bfloat16, device='cuda')
delta = torch.empty(B, H, N, device='cuda')
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    lib.vitae_sdpa_mfma_fwd_bf16in(P(q16), P(o), P(o16), P(lse), B, N, H, hd, st)
    lib.vitae_sdpa_mfma_bwd_bf16in(P(q16), P(o), P(do), P(lse), None, P(d16), None, P(delta), B, N, H, hd, st)
torch.cuda.synchronize()
