"""AdamW / grad-norm kernel timing at the ViT-B contrastive arena size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd import _abi
lib, C = _abi.lib, _abi.CONSTS
n = 132_000_000
p, g, m, v = (torch.randn(n, device='cuda') * 0.02 for _ in range(4))
v.abs_()
sh = torch.empty(n, dtype=torch.bfloat16, device='cuda')
hp = torch.zeros(C['VITAE_HP_COUNT'], device='cuda')
hp[C['VITAE_HP_LR']] = 1e-4; hp[C['VITAE_HP_BETA1']] = 0.9; hp[C['VITAE_HP_BETA2']] = 0.95; hp[C['VITAE_HP_EPS']] = 1e-8
hp[C['VITAE_HP_BC1']] = 0.1; hp[C['VITAE_HP_BC2']] = 0.05; hp[C['VITAE_HP_GRAD_MUL']] = 1.0
acc = torch.zeros(C['VITAE_ACC_COUNT'], dtype=torch.float64, device='cuda'); gn = torch.zeros(1, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def t(fn, k=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3
us = t(lambda: lib.vitae_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr(), n, hp.data_ptr(), None, 0.05, st))
print(f'adamw, fp32 moments {us:.1f} us  {n * 30 / us / 1e6:.2f} TB/s (30 B per parameter)')
m16, v16 = m.to(torch.bfloat16), v.to(torch.bfloat16)
us = t(lambda: lib.vitae_adamw_step_s16(p.data_ptr(), g.data_ptr(), 0, m16.data_ptr(), v16.data_ptr(), sh.data_ptr(), n, hp.data_ptr(), None, 0.05, st))
print(f'adamw, bf16 moments {us:.1f} us  {n * 22 / us / 1e6:.2f} TB/s (22 B per parameter)')
g16 = g.to(torch.bfloat16)
us = t(lambda: lib.vitae_adamw_step_s16(p.data_ptr(), g16.data_ptr(), 1, m16.data_ptr(), v16.data_ptr(), sh.data_ptr(), n, hp.data_ptr(), None, 0.05, st))
print(f'adamw, bf16 moments + bf16 gradients {us:.1f} us  {n * 20 / us / 1e6:.2f} TB/s (20 B per parameter)')
us = t(lambda: lib.vitae_grad_sqnorm(g.data_ptr(), n, acc.data_ptr(), gn.data_ptr(), st))
print(f'gradnorm {us:.1f} us  {n * 4 / us / 1e6:.2f} TB/s')
