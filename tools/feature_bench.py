"""Throughput of encoder-only feature extraction (VisionTransformer3D.forward_features, ViT-B/16, 96^3 x 4ch, unmasked:
217 tokens per volume) — SURVEY §8(f) row 1.  Eager launches (the path utils/feature_extraction.generate_features takes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_ae_plus_plus_amd.model.vit import VisionTransformer3D

for precision in ('bf16', 'fp32'):
    for B in (4, 16):
        m = VisionTransformer3D(volume_size=96, patch_size=16, in_chans=4, num_classes=2, global_pool=True, precision=precision).cuda().eval()
        x = torch.randn(B, 4, 96, 96, 96, device='cuda')
        with torch.no_grad():
            for _ in range(3):
                m.forward_features(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                f = m.forward_features(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        print(f'{precision} B={B}: {dt * 1e3:.2f} ms / batch, {B / dt:.0f} volumes/s')
