"""Step time with and without the optimiser (update=False micro-steps run forward + backward only): what grad-norm + AdamW
cost in the step, overlap and contention included."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mae_ref as R
from vit_ae_plus_plus_amd.model import vit_autoenc as VA
from vit_ae_plus_plus_amd.optim import FusedAdamW

dev = torch.device('cuda', 0)
margs = argparse.Namespace(use_imagenet=False, perceptual_weight=0)
model = VA.contr_mae_vit_base_patch16(volume_size=96, in_chans=4, patch_size=16, args=margs, precision='bf16').to(dev).train()
eng = model._ensure_engine(dev)
opt = FusedAdamW(model, lr=1e-4, weight_decay=0.05, betas=(0.9, 0.95)); _ = opt.engine
eng.set_loss_weights(0.01, 0.001, 1, 1)
B = 4
g = torch.Generator(device='cuda').manual_seed(1)
v1, v2 = torch.randn(B, 4, 96, 96, 96, device=dev, generator=g), torch.randn(B, 4, 96, 96, 96, device=dev, generator=g)
for update in (True, False, True, False):
    runner = model._step_runner(B, 0.75, update, False, True)
    for i in range(6 + 60):
        if i == 6:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        runner.load(v1, v2)
        if update:
            eng.optimizer_hparams(lr=1e-4)
        runner.run()
    torch.cuda.synchronize()
    print(f'update={update}: {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/step')
