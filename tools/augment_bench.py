"""Time of producing both views of a raw batch on the GPU (utils.augment.augmented_views), per kernel group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd.utils.augment import Compose, RandomAffine, RandomGamma, RandomNoise, augmented_views
from vit_ae_plus_plus_amd.utils.input_pipeline import normalize_data

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
raw = torch.randn(B, 4, 96, 96, 96, device='cuda') * 40 + 100
ra, rn, rg = RandomAffine(), RandomNoise(std=0.1), RandomGamma()
tf = Compose([ra, rn, rg])


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


mb = raw.numel() * 4 / 1e6
mats = RandomAffine.matrices(*ra.get_params(B), raw.shape[2:])
noise = torch.randn_like(raw)
print(f'batch {B}: {mb:.0f} MB per view')
print(f'  affine (minmax + resample)   {timed(lambda: ra.apply(raw, mats)):8.1f} us')
print(f'  noise + gamma (fused)        {timed(lambda: Compose([rn, rg])(raw)):8.1f} us   (of which torch.randn {timed(lambda: torch.randn_like(raw)):.1f})')
print(f'  z-score normalisation        {timed(lambda: normalize_data(raw, True)):8.1f} us')
t = timed(lambda: augmented_views(raw, tf, True))
print(f'  both views, end to end       {t:8.1f} us  = {B / t * 1e6:.0f} volumes/s')
