"""Per-kernel timing of the loss chain at the bench shape (B=4, 4ch, 96^3, p=16): events around 20 back-to-back launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd import _abi
from vit_ae_plus_plus_amd.engine import gaussian_taps_host

lib, C = _abi.lib, _abi.CONSTS
B, Cc, vol, p = int(os.environ.get('LB_BATCH', '4')), 4, (96, 96, 96), int(os.environ.get('LB_PATCH', '16'))
V = vol[0] * vol[1] * vol[2]
L, P = (vol[0] // p) ** 3, p ** 3 * Cc
g = torch.Generator(device='cuda').manual_seed(0)
imgs = torch.randn(B, Cc, *vol, device='cuda', generator=g)
predfull = torch.randn(B, L + 1, P, device='cuda', generator=g) * 0.5
mask = (torch.rand(B, L, device='cuda', generator=g) < 0.75).float()
hp = torch.zeros(C['VITAE_HP_COUNT'], device='cuda'); hp[C['VITAE_HP_G_RECON']] = 1; hp[C['VITAE_HP_G_EDGE']] = 0.01
acc = torch.zeros(C['VITAE_ACC_COUNT'], dtype=torch.float64, device='cuda')
pv, tmp, bl = torch.empty_like(imgs), torch.empty_like(imgs), torch.empty_like(imgs)
et, ep = torch.empty(B, *vol, device='cuda'), torch.empty(B, *vol, device='cuda')
dpred = torch.zeros(B, L + 1, P, device='cuda'); d16 = torch.zeros(B, L + 1, P, dtype=torch.bfloat16, device='cuda')
taps = gaussian_taps_host(2.0)
st = torch.cuda.current_stream().cuda_stream
pp, pbs = predfull.data_ptr() + P * 4, (L + 1) * P
msum = float(mask.sum())
ops = {
    'recon_fwd': lambda: lib.vitae_recon_loss_fwd(pp, pbs, imgs.data_ptr(), mask.data_ptr(), acc.data_ptr(), B, Cc, *vol, p, st),
    'unpatchify': lambda: lib.vitae_unpatchify(pp, pbs, pv.data_ptr(), B, Cc, *vol, p, st),
    'blur(2 kernels)': lambda: lib.vitae_gauss_blur_fwd(imgs.data_ptr(), tmp.data_ptr(), bl.data_ptr(), taps.ctypes.data, len(taps), B * Cc, *vol, st),
    'sobel_tgt': lambda: lib.vitae_sobel_edge_fwd(bl.data_ptr(), et.data_ptr(), None, None, B, Cc, *vol, st),
    'sobel_pred+mse': lambda: lib.vitae_sobel_edge_fwd(pv.data_ptr(), ep.data_ptr(), et.data_ptr(), acc.data_ptr(), B, Cc, *vol, st),
    'loss_fwd_fused': lambda: lib.vitae_loss_fwd_fused(pp, pbs, imgs.data_ptr(), mask.data_ptr(), et.data_ptr(), pv.data_ptr(), ep.data_ptr(), acc.data_ptr(), B, Cc, *vol, p, st),
    'loss_bwd_fused': lambda: lib.vitae_loss_bwd_fused(pp, pv.data_ptr(), imgs.data_ptr(), mask.data_ptr(), ep.data_ptr(), et.data_ptr(), hp.data_ptr(), None,
                                                      dpred.data_ptr() + P * 4, d16.data_ptr() + P * 2, None, pbs, msum, B, Cc, *vol, p, st),
    'target_edge (one pass)': lambda: lib.vitae_target_edge(imgs.data_ptr(), et.data_ptr(), taps.ctypes.data, len(taps), B, Cc, *vol, st),
    'loss_fwd_bwd (one pass)': lambda: lib.vitae_loss_fwd_bwd(pp, pbs, imgs.data_ptr(), mask.data_ptr(), et.data_ptr(), hp.data_ptr(),
                                                              dpred.data_ptr() + P * 4, d16.data_ptr() + P * 2, None, acc.data_ptr(), msum, B, Cc, *vol, p, st),
    'loss_fwd_bwd (bf16 gradient only: the step)': lambda: lib.vitae_loss_fwd_bwd(pp, pbs, imgs.data_ptr(), mask.data_ptr(), et.data_ptr(), hp.data_ptr(),
                                                              None, d16.data_ptr() + P * 2, None, acc.data_ptr(), msum, B, Cc, *vol, p, st),
}
if os.environ.get('LB_ONLY'):                     # e.g. LB_ONLY="one pass,gradient only"
    ops = {k: v for k, v in ops.items() if any(w in k for w in os.environ['LB_ONLY'].split(','))}
for name, fn in ops.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record(); torch.cuda.synchronize()
    print(f'{name:44s} {a.elapsed_time(b) / 20 * 1e3:8.1f} us')
