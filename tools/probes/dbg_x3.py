import sys, os, argparse
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import torch, numpy as np
from conftest import load_golden
from oracle import mae_ref as R
import test_gpu_model as T
from vit_ae_plus_plus_amd.utils.train_one_epoch import compute_contrastive_loss
cfg = R.vit_base_cfg(volume_size=(96, 96, 96), patch_size=8, in_chans=4, contrastive=True)
g = load_golden('vitb_p8.npz')
model = T.build(cfg, R.init_state_dict(cfg, seed=0), precision='fp32x3')
model.train(True)
v1, v2 = R.synthetic_views((1, cfg.in_chans, *cfg.volume_size), seed=1234)
n1, n2 = R.masking_noise(1, cfg.num_patches, seed=4321)
model.set_masking_noise(n1, n2)
loss, pred, mask, p1, p2, z1, z2 = model(view1=v1.cuda(), view2=v2.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
contr = compute_contrastive_loss(argparse.Namespace(contr_weight=0.001), None, p1, p2, z1, z2)
(loss[0] + contr).backward()
named = dict(model.named_parameters())
for k, refn in zip(list(g['grad_names']), g['grad_norms']):
    gotn = float(named[str(k)].grad.double().norm())
    if abs(gotn - refn) > 2e-3 * refn and refn > 1e-8:
        print(f'{str(k):40s} got {gotn:.6e} ref {refn:.6e}')
print('done')
