import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mae_ref as R
from vit_ae_plus_plus_amd.model import vit_autoenc as VA
from vit_ae_plus_plus_amd.utils.train_one_epoch import compute_contrastive_loss
cfg = R.vit_base_cfg(volume_size=(96,)*3, patch_size=16, in_chans=4, contrastive=True)
sd = R.init_state_dict(cfg, seed=0)
m = VA.contr_mae_vit_base_patch16(volume_size=96, in_chans=4, patch_size=16, args=argparse.Namespace(use_imagenet=False, perceptual_weight=0), precision=sys.argv[1] if len(sys.argv) > 1 else 'bf16')
m.load_state_dict(sd); m = m.cuda().train()
v1, v2 = R.synthetic_views((4, 4, 96, 96, 96), seed=1234)
n1, n2 = R.masking_noise(4, cfg.num_patches, seed=4321)
m.set_masking_noise(n1, n2)
loss, pred, mask, p1, p2, z1, z2 = m(view1=v1.cuda(), view2=v2.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
c = compute_contrastive_loss(argparse.Namespace(contr_weight=0.001), None, p1, p2, z1, z2)
print('losses', [float(x) for x in loss], float(c))
(loss[0] + c).backward()
bad = 0
for k, p in m.named_parameters():
    if p.requires_grad:
        g = p.grad
        if not torch.isfinite(g).all():
            bad += 1
            print('NONFINITE', k, int((~torch.isfinite(g)).sum()), g.numel())
print('bad tensors', bad)
eng = m.engine
for k in ['dpredfull', 'ddn', 'decdx', 'de', 'dlatent', 'encdx', 'dtok', 'dG', 'edge_p', 'edge_t']:
    t = eng.buf[k]
    print(k, 'finite' if torch.isfinite(t).all() else f'NONFINITE {int((~torch.isfinite(t)).sum())}/{t.numel()}', float(t[torch.isfinite(t)].abs().max()))

print('---- fused path')
from vit_ae_plus_plus_amd.optim import FusedAdamW
opt = FusedAdamW(m, lr=1e-4, weight_decay=0.05)
_ = opt.engine
eng.set_loss_weights(0.01, 0.001, 1, 1)
for use_graph in (False, True):
    r = m._step_runner(4, 0.75, True, False, use_graph)
    for step in range(3):
        r.load(v1.cuda(), v2.cuda())
        eng.optimizer_hparams(lr=1e-4)
        r.run()
        torch.cuda.synchronize()
        g = eng.grads
        nf = int((~torch.isfinite(g)).sum())
        print('graph' if use_graph else 'eager', step, 'losses', [round(x, 5) for x in eng.losses.cpu().tolist()[:6]], 'nonfinite grads', nf,
              'max|g|', float(g[torch.isfinite(g)].abs().max()), 'params finite', bool(torch.isfinite(eng.params).all()),
              'shadow finite', bool(torch.isfinite(eng.params16.float()).all()))
        if nf:
            for k, (o, shp) in eng.layout.items():
                t = eng.g[k]
                if not torch.isfinite(t).all():
                    print('   NONFINITE', k, int((~torch.isfinite(t)).sum()), t.numel())
print('---- which tensors are huge')
for k, (o, shp) in eng.layout.items():
    t = eng.g[k]
    mx = float(t.abs().max())
    if mx > 10:
        idx = (t.abs() > 10).nonzero()
        print('   HUGE', k, tuple(shp), 'count', idx.shape[0], 'first', idx[:3].tolist(), 'max', mx)
g = eng.grads
i = int(g.abs().argmax()); print('argmax index', i, 'value', float(g[i]), 'n_total', eng.n_total, 'tok_off', eng.tok_off, 'vec_off', eng.vec_off)
prev = None
for k, (o, shp) in eng.layout.items():
    n = 1
    for s_ in shp: n *= s_
    if o <= i < o + n: print('inside', k, shp, 'local', i - o)
    if prev is not None and prev[1] != o: print('GAP between', prev[0], 'end', prev[1], 'and', k, 'start', o)
    prev = (k, o + n)
print('last end', prev, 'sum check', float(eng.losses[5]))
