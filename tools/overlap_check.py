"""Race check for the side-stream overlaps of the fused step: the same 25 optimisation steps (ViT-B contrastive, 96^3 x 4ch,
batch 4, bf16, graph replay) with and without the predictor / optimiser side branches must land on the same weights up
to the round-off of atomically accumulated reductions.  Each mode runs in its own process (the knobs are read at import)."""
import json, os, subprocess, sys

CHILD = r'''
import os, sys, json, argparse, torch
sys.path.insert(0, os.getcwd())
from oracle import mae_ref as R
from vit_ae_plus_plus_amd.model import vit_autoenc as VA
from vit_ae_plus_plus_amd.optim import FusedAdamW
cfg = R.vit_base_cfg(volume_size=(96, 96, 96), patch_size=16, in_chans=4, contrastive=True)
sd = R.init_state_dict(cfg, seed=0)
m = VA.contr_mae_vit_base_patch16(volume_size=96, in_chans=4, patch_size=16, args=argparse.Namespace(use_imagenet=False, perceptual_weight=0), precision='bf16')
m.load_state_dict(sd); m = m.cuda().train()
eng = m._ensure_engine(torch.device('cuda', 0))
opt = FusedAdamW(m, lr=1e-4, weight_decay=0.05); _ = opt.engine
eng.set_loss_weights(0.01, 0.001, 1, 1)
B = 4
runner = m._step_runner(B, 0.75, True, False, True)
g = torch.Generator(device='cuda').manual_seed(5)
batches = [(torch.randn(B, 4, 96, 96, 96, device='cuda', generator=g), torch.randn(B, 4, 96, 96, 96, device='cuda', generator=g)) for _ in range(3)]
for i in range(25):
    v1, v2 = batches[i % 3]
    m.set_masking_noise(*R.masking_noise(B, cfg.num_patches, seed=100 + i))
    runner.load(v1, v2); eng.optimizer_hparams(lr=1e-4); runner.run()
torch.cuda.synchronize()
p = eng.params.double()
print(json.dumps({'sum': float(p.sum()), 'abs': float(p.abs().sum()), 'sq': float((p * p).sum()), 'loss': eng.losses.cpu().tolist()[:6],
                  'probe': eng.params[::1000003].cpu().tolist()[:40]}))
'''

def run(env):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, '-c', CHILD], env=e, capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if not line:
        raise SystemExit(out.stdout[-2000:] + out.stderr[-2000:])
    return json.loads(line[-1])

modes = {'serial': {'VITAE_OPT_IN_BACKWARD': '0', 'VITAE_PREDICTOR_SIDE': '0'}, 'overlap': {}, 'overlap_again': {}}
res = {k: run(v) for k, v in modes.items()}
ref = res['serial']
for k, r in res.items():
    dp = max(abs(a - b) for a, b in zip(r['probe'], ref['probe']))
    print(k, 'loss', [round(x, 6) for x in r['loss']], 'rel d|p|', abs(r['abs'] - ref['abs']) / ref['abs'], 'max probe diff', dp)
