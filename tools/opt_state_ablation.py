"""What does a 16-bit AdamW moment do to a training trajectory?  (measure, do not assume)

CPU experiment on the oracle (plain PyTorch fp32, oracle/train_ref.py): torch.optim.AdamW as the reference drives it, but after every
step the stored exp_avg (m) and / or exp_avg_sq (v) are rounded to bf16 — exactly what a kernel that keeps the moment in bf16 and
computes in fp32 does.  Two workloads:
    pins    the pinned B = 4 ViT-B trajectory of tests/golden/vitb_b4.npz (three AdamW steps at lr 1e-4 + the losses of a fourth batch,
            from the reference's own model): worst relative error of [total, raw edge, recon, contrastive] per variant
    tiny    BASELINE config 1's model (64^3 x 1ch, D = 128, depth 2) for N steps at a production learning rate on 8 recurring volumes:
            the loss curve of each variant against the fp32-state curve (relative difference per step, and the spread between two
            fp32-state runs that differ only in the masking seed, as the yardstick of what "different" means for this loss)
    python tools/opt_state_ablation.py [pins] [tiny] [steps=60] [lr=1e-3]
Nothing here is imported by the product."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import mae_ref as R
from oracle import train_ref as T
from oracle.gen_golden import VITB

torch.set_num_threads(int(os.environ.get('ABL_THREADS', '8')))
args = dict(a.split('=') for a in sys.argv[1:] if '=' in a)
what = [a for a in sys.argv[1:] if '=' not in a] or ['pins', 'tiny']
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
VARIANTS = {'fp32 state': (), 'm bf16': ('exp_avg',), 'v bf16': ('exp_avg_sq',), 'm + v bf16': ('exp_avg', 'exp_avg_sq'),
            'm + v + matrix gradients bf16': ('exp_avg', 'exp_avg_sq', 'GRAD')}     # GRAD: the gradients of the matrices rounded before the step


def round_state(tr, keys):
    for st in tr.optimizer.state.values():
        for k in keys:
            if k != 'GRAD':
                st[k].copy_(bf(st[k]))


def install_grad_rounding(tr, keys):
    """'GRAD': round the gradient of every matrix to bf16 right before optimizer.step() (what a weight-gradient epilogue that stores bf16 only does)"""
    if 'GRAD' not in keys:
        return
    orig = tr.optimizer.step

    def step(*a, **k):
        for p in tr.params.values():
            if p.grad is not None and p.ndim >= 2:
                p.grad.copy_(bf(p.grad))
        return orig(*a, **k)
    tr.optimizer.step = step


def pins():
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'vitb_b4.npz'))
    B, steps, lr, wd, mask_ratio, edge_w, contr_w = [float(v) for v in g['hp']]
    B, steps = int(B), int(steps)
    cfg = R.vit_base_cfg(contrastive=True, **VITB)
    for name, keys in VARIANTS.items():
        tr = T.RefTrainer(cfg, R.init_state_dict(cfg, seed=0), lr=lr, weight_decay=wd)
        install_grad_rounding(tr, keys)
        worst, t0 = [0.0] * 4, time.time()
        for it in range(steps + 1):
            v1, v2 = R.synthetic_views((4, 4, 96, 96, 96), seed=1234 + it)
            n1, n2 = R.masking_noise(4, cfg.num_patches, seed=4321 + it)
            terms, _, _ = tr.step(v1, v2, n1, n2, lr=lr, mask_ratio=mask_ratio, edge_map_weight=edge_w, contr_weight=contr_w)
            round_state(tr, keys)
            got = [terms['loss'] - terms['contr_loss'], terms['edge_map_loss'], terms['reconstruction_loss'], terms['contr_loss']]
            want = [g['losses'][it][i] for i in (0, 1, 2, 4)]
            worst = [max(w, abs(a - b) / (abs(b) + 1e-12)) for w, a, b in zip(worst, got, want)]
        print(f'pins  {name:12s} worst rel err [total, raw edge, recon, contr] = ' + ' '.join(f'{w:9.2e}' for w in worst) + f'   ({time.time() - t0:.0f} s)', flush=True)


def tiny():
    steps, lr = int(args.get('steps', 60)), float(args.get('lr', 1e-3))
    cfg = R.RefConfig(volume_size=(64, 64, 64), patch_size=16, in_chans=1, embed_dim=128, depth=2, num_heads=4,
                      decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=4, contrastive=True)
    vols = [R.synthetic_views((4, 1, 64, 64, 64), seed=100 + i) for i in range(2)]

    def curve(keys, noise_seed):
        tr = T.RefTrainer(cfg, R.init_state_dict(cfg, seed=0), lr=lr, weight_decay=0.05)
        install_grad_rounding(tr, keys)
        out = []
        for it in range(steps):
            v1, v2 = vols[it % len(vols)]
            n1, n2 = R.masking_noise(4, cfg.num_patches, seed=noise_seed + it)
            terms, _, _ = tr.step(v1, v2, n1, n2, lr=lr, mask_ratio=0.75, edge_map_weight=0.01, contr_weight=0.001)
            round_state(tr, keys)
            out.append(terms['loss'])
        return np.array(out)

    base = curve((), 7000)
    other = curve((), 9000)
    print(f'tiny  {steps} steps at lr {lr:g}: loss {base[0]:.4f} -> {base[-1]:.4f}; step-to-step |change| median {np.median(np.abs(np.diff(base)) / base[1:]):.2e}; '
          f'another masking seed differs by median {np.median(np.abs(other - base) / base):.2e} (max {np.max(np.abs(other - base) / base):.2e})', flush=True)
    for name, keys in VARIANTS.items():
        if not keys:
            continue
        c = curve(keys, 7000)
        rel = np.abs(c - base) / base
        print(f'tiny  {name:12s} vs fp32 state: rel diff of the loss per step: median {np.median(rel):.2e}, max {rel.max():.2e}, last step {rel[-1]:.2e}', flush=True)


if 'pins' in what:
    pins()
if 'tiny' in what:
    tiny()
