"""Step time of the other BASELINE configurations (parity cases, not bench lines): config 4 = ViT-L/16 MAE on 128^3 x 4ch,
config 5 = ViT-B/16 on EGD-shape 192 x 192 x 32 x 1ch volumes.  Same fused step / graph replay as bench.py."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mae_ref as R
from vit_ae_plus_plus_amd.model import vit_autoenc as VA
from vit_ae_plus_plus_amd.optim import FusedAdamW

CASES = {
    'config4 ViT-L/16 MAE 128^3x4ch': dict(ctor='mae_vit_large_patch16', vol=128, ch=4, contr=False, gflop_per_vol=406.8, batch=4),
    'config5 ViT-B/16 MAE 192x192x32x1ch': dict(ctor='mae_vit_base_patch16', vol=(192, 192, 32), ch=1, contr=False, gflop_per_vol=94.8, batch=4),
}
margs = argparse.Namespace(use_imagenet=False, perceptual_weight=0)
dev = torch.device('cuda', 0)
for name, c in CASES.items():
    model = getattr(VA, c['ctor'])(volume_size=c['vol'], in_chans=c['ch'], patch_size=16, args=margs, precision='bf16').to(dev).train()
    eng = model._ensure_engine(dev)
    opt = FusedAdamW(model, lr=1e-4, weight_decay=0.05, betas=(0.9, 0.95)); _ = opt.engine
    eng.set_loss_weights(0.01, 0.0, 1, 1)
    B = c['batch']
    vol = c['vol'] if isinstance(c['vol'], tuple) else (c['vol'],) * 3
    g = torch.Generator(device='cuda').manual_seed(1)
    batches = [torch.randn(B, c['ch'], *vol, device=dev, generator=g) for _ in range(3)]
    runner = model._step_runner(B, 0.75, True, False, True)
    warm, steps = 2 * len(batches) + 2, 15     # every resident batch twice first: in-place graphs are captured on second sight
    for i in range(warm + steps):
        if i == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        runner.load(batches[i % 3], None); eng.optimizer_hparams(lr=1e-4); runner.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n_par = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print(f'{name}: batch {B}, {n_par / 1e6:.1f} M params, {dt * 1e3:.2f} ms/step, {B / dt:.1f} volumes/s, '
          f'{B * c["gflop_per_vol"] / dt / 1e3:.1f} TFLOP/s of reference-formulation work, act16={eng.act16}, losses {[round(x, 4) for x in eng.losses.cpu().tolist()[:3]]}')
    del model, eng, opt, runner, batches
    torch.cuda.empty_cache()
