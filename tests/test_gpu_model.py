"""GPU parity of the whole path through the drop-in Python surface (C ABI underneath):
micro model against the fixtures generated from the reference, config-1 epoch against the
reference loop's stats, ViT-B (config-2 shape) against the reference pins and the live oracle."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden
from oracle import mae_ref as R
from oracle import train_ref as T
from oracle.gen_golden import MICRO, TINY, VITB


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol, atol):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def build(cfg: R.RefConfig, sd, precision='fp32'):
    from functools import partial
    from vit_ae_plus_plus_amd.model import vit_autoenc as VA
    args = argparse.Namespace(use_imagenet=False, perceptual_weight=0)
    cls = VA.ContrastiveMAEViT if cfg.contrastive else VA.MaskedAutoencoderViT
    vol = cfg.volume_size if len(set(cfg.volume_size)) > 1 else cfg.volume_size[0]
    m = cls(volume_size=vol, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim,
            depth=cfg.depth, num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim,
            decoder_depth=cfg.decoder_depth, decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio,
            norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps), args=args, precision=precision)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    return m.cuda()


@pytest.mark.parametrize('name,contrastive', [('micro.npz', True), ('micro_mae.npz', False)])
def test_micro_vs_reference_fixture(name, contrastive):
    from vit_ae_plus_plus_amd.utils.train_one_epoch import compute_contrastive_loss
    g = load_golden(name)
    cfg = R.RefConfig(contrastive=contrastive, **MICRO)
    sd = {k[3:]: t(g[k]) for k in g.files if k.startswith('sd/')}
    model = build(cfg, sd)
    model.train(True)
    mask_ratio, edge_w, contr_w = [float(v) for v in g['hp']]
    v1, v2 = t(g['view1']).cuda(), t(g['view2']).cuda()
    if contrastive:
        model.set_masking_noise(t(g['noise1']), t(g['noise2']))
        loss, pred, mask, p1, p2, z1, z2 = model(view1=v1, view2=v2, mask_ratio=mask_ratio, edge_map_weight=edge_w)
        contr = compute_contrastive_loss(argparse.Namespace(contr_weight=contr_w), None, p1, p2, z1, z2)
    else:
        model.set_masking_noise(t(g['noise1']))
        loss, pred, mask = model(v1, mask_ratio=mask_ratio, edge_map_weight=edge_w)
        contr = torch.zeros((), device='cuda')
    assert torch.equal(mask.cpu(), t(g['mask']))
    close(torch.stack(loss), g['losses'], 2e-5, 1e-6)
    close(contr, g['contr_loss'], 2e-4, 1e-8)
    close(pred, g['pred'], 1e-3, 2e-5)
    eng = model.engine
    close(eng.buf['latent'][:eng.B * eng.Ne].reshape(eng.B, eng.Ne, -1), g['t/latent'], 1e-3, 2e-5)
    close(eng.buf['decx'][0].reshape(eng.B, eng.Nd, -1), g['t/decoder_in'], 1e-3, 2e-5)
    close(eng.buf['blurred'].reshape(g['t/blurred'].shape), g['t/blurred'], 1e-4, 1e-6)
    close(eng.buf['edge_t'].reshape(g['t/edge_target'].shape), g['t/edge_target'], 1e-4, 1e-5)
    close(eng.buf['edge_p'].reshape(g['t/edge_pred'].shape), g['t/edge_pred'], 1e-3, 1e-4)
    if contrastive:
        close(p1, g['p1'], 1e-3, 2e-5)
        close(p2, g['p2'], 1e-3, 2e-5)
        close(z1, g['z1'], 1e-3, 2e-5)
        close(eng.buf['latent'][eng.R:].reshape(eng.B, eng.Ne, -1), g['t2/latent'], 1e-3, 2e-5)
        close(model.predictor[1].running_mean, g['bn_running_mean'], 1e-4, 1e-6)
        close(model.predictor[1].running_var, g['bn_running_var'], 1e-4, 1e-6)
        assert int(model.predictor[1].num_batches_tracked) == 2
    (loss[0] + contr).backward()
    n = 0
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        ref = g['grad/' + k]
        assert p.grad is not None, k
        scale = max(1e-12, float(np.abs(ref).max()))
        close(p.grad.cpu().numpy() / scale, ref / scale, 2e-3, 1e-4)
        n += 1
    assert n == sum(1 for k in g.files if k.startswith('grad/'))


def _tiny_loader(cfg, iters=2, B=4):
    data, noises = [], []
    for it in range(iters):
        v1, v2 = R.synthetic_views((B, 1, 64, 64, 64), seed=1234 + it)
        data.append((v1, v2, torch.zeros(B)))
        noises.extend(R.masking_noise(B, cfg.num_patches, seed=4321 + it))
    return data, noises


@pytest.mark.parametrize('route', ['fused_graph', 'fused_eager', 'autograd'])
def test_config1_epoch_vs_reference_loop(route):
    """BASELINE config 1 through train_one_stage_epoch: stats of the reference's own loop (fixture)."""
    from vit_ae_plus_plus_amd.utils import misc
    from vit_ae_plus_plus_amd.utils.train_one_epoch import train_one_stage_epoch
    g = load_golden('tiny_epoch.npz')
    lr, wd, edge_w, contr_w, mask_ratio, warm, epochs = [float(v) for v in g['hp']]
    cfg = R.RefConfig(contrastive=True, **TINY)
    model = build(cfg, R.init_state_dict(cfg, seed=0))
    data, noises = _tiny_loader(cfg)
    model.set_masking_noise(*noises)
    args = argparse.Namespace(accum_iter=1, mask_ratio=mask_ratio, contr_weight=contr_w, lr=lr, min_lr=0.0,
                              warmup_epochs=warm, epochs=epochs, hip_graph=(route == 'fused_graph'),
                              no_fused_step=(route == 'autograd'))
    groups = R.param_groups(dict(model.named_parameters()), wd)
    named = dict(model.named_parameters())
    opt = torch.optim.AdamW([{'params': [named[n] for n in gr['names']], 'weight_decay': gr['weight_decay']}
                             for gr in groups], lr=lr, betas=(0.9, 0.95))
    stats = train_one_stage_epoch(model, data, opt, torch.device('cuda'), int(g['epoch']), misc.NativeScalerWithGradNormCount(),
                                  log_writer=None, args=args, edge_map_weight=edge_w)
    assert set(stats) == {'lr', 'edge_map_loss', 'reconstruction_loss', 'perceptual_loss', 'contr_loss', 'loss'}
    for k, v in stats.items():
        close(v, g['stat/' + k], 1e-4, 1e-7)
    fin = model.state_dict()
    for k in g.files:
        if k.startswith('norm/'):
            # biases: Adam amplifies atomic-order round-off where the true gradient is ~0 (e.g. key biases)
            close(float(fin[k[5:]].double().norm()), g[k], 5e-4 if k.endswith('.bias') else 2e-5, 1e-7)
    close(fin['cls_token'], g['final/cls_token'], 1e-3, 1e-6)
    close(fin['predictor.1.running_var'], g['final/predictor.1.running_var'], 1e-3, 1e-6)
    osd = opt.state_dict()
    assert len(osd['state']) == len(named) - 0 - sum(1 for p in named.values() if not p.requires_grad)


@pytest.mark.parametrize('tag', ['mae', 'contr'])
def test_vitb_config2_vs_reference_pins(tag):
    """ViT-B/16, 96^3 x 4ch, B=2: loss scalars, mask sums, pred samples and per-parameter gradient
    norms pinned from the reference (tests/golden/vitb.npz); 1e-4 relative on the losses is the
    tolerance BASELINE.json's north_star states."""
    from vit_ae_plus_plus_amd.utils.train_one_epoch import compute_contrastive_loss
    g = load_golden('vitb.npz')
    contrastive = tag == 'contr'
    cfg = R.vit_base_cfg(contrastive=contrastive, **VITB)
    model = build(cfg, R.init_state_dict(cfg, seed=0))
    model.train(True)
    v1, v2 = R.synthetic_views((2, 4, 96, 96, 96), seed=1234)
    n1, n2 = R.masking_noise(2, cfg.num_patches, seed=4321)
    if contrastive:
        model.set_masking_noise(n1, n2)
        loss, pred, mask, p1, p2, z1, z2 = model(view1=v1.cuda(), view2=v2.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
        contr = compute_contrastive_loss(argparse.Namespace(contr_weight=0.001), None, p1, p2, z1, z2)
        close(p1[::11, ::97], g['contr/p1_slice'], 2e-3, 1e-4)
    else:
        model.set_masking_noise(n1)
        loss, pred, mask = model(v1.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
        contr = torch.zeros((), device='cuda')
    close(torch.stack(loss), g[f'{tag}/losses'], 1e-4, 1e-7)
    close(contr, g[f'{tag}/contr_loss'], 1e-3, 1e-9)
    assert torch.equal(mask.sum(1).cpu(), t(g[f'{tag}/mask_sum']))
    close(pred[:, ::37, ::1021], g[f'{tag}/pred_slice'], 2e-3, 1e-4)
    (loss[0] + contr).backward()
    names, norms = list(g[f'{tag}/grad_names']), g[f'{tag}/grad_norms']
    named = dict(model.named_parameters())
    for k, ref in zip(names, norms):
        got = float(named[str(k)].grad.double().norm())
        assert abs(got - ref) <= 2e-3 * ref + 1e-9, (k, got, ref)


def test_bf16_mode_close_to_fp32_reference():
    """bf16-MFMA mode (fp32 storage / accumulation): loss within 1e-2 relative of the fp32 pins;
    the precise figure is reported by bench.py (north_star asks for 1e-4 on the fp32 parity mode)."""
    g = load_golden('vitb.npz')
    cfg = R.vit_base_cfg(contrastive=False, **VITB)
    model = build(cfg, R.init_state_dict(cfg, seed=0), precision='bf16')
    v1, _ = R.synthetic_views((2, 4, 96, 96, 96), seed=1234)
    n1, _ = R.masking_noise(2, cfg.num_patches, seed=4321)
    model.set_masking_noise(n1)
    loss, pred, mask = model(v1.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
    close(torch.stack(loss)[:3], g['mae/losses'][:3], 1e-2, 1e-6)
    loss[0].backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.requires_grad)


def test_cpu_input_fails_loudly():
    from vit_ae_plus_plus_amd._abi import VitaeError
    cfg = R.RefConfig(contrastive=False, **MICRO)
    model = build(cfg, R.init_state_dict(cfg, seed=0))
    with pytest.raises(VitaeError):
        model(torch.zeros(1, 2, 16, 16, 16))


def test_three_steps_track_oracle_training():
    """Three AdamW steps (fused graph route) on the micro model follow the fp32 oracle trainer."""
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    cfg = R.RefConfig(contrastive=True, **MICRO)
    sd = R.init_state_dict(cfg, seed=5)
    model = build(cfg, sd)
    tr = T.RefTrainer(cfg, sd, lr=1e-3, weight_decay=0.05)
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05)
    model._ensure_engine(torch.device('cuda', 0))
    eng = opt.engine
    eng.set_loss_weights(0.01, 0.001, 1)
    B = 2
    for step in range(3):
        v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=100 + step)
        n1, n2 = R.masking_noise(B, cfg.num_patches, seed=200 + step)
        terms, norm, _ = tr.step(v1, v2, n1, n2, lr=1e-3, mask_ratio=0.75, edge_map_weight=0.01, contr_weight=0.001)
        model.set_masking_noise(n1, n2)
        runner = model._step_runner(B, 0.75, True, False, True)
        runner.load(v1, v2)
        eng.optimizer_hparams(lr=1e-3)
        runner.run()
        got = eng.losses.cpu().tolist()
        close(got[0] + got[4], terms['loss'], 2e-4, 1e-7)
        close(got[2], terms['reconstruction_loss'], 2e-4, 1e-7)
        close(got[5], float(norm), 2e-3, 1e-7)
    ref_sd = tr.state_dict()
    for k, v in model.state_dict().items():
        if not v.is_floating_point() or not R.is_trainable(k):
            continue
        # compare the 3-step UPDATE per tensor.  Adam's m/(sqrt(v)+eps) turns round-off into +-lr steps
        # wherever the true gradient is ~0 (e.g. the key bias, whose gradient is analytically zero), so
        # elementwise equality is ill-posed there; the update's relative L2 error is the stable measure.
        if k.endswith('qkv.bias'):
            # q | k | v thirds on their own: only the KEY bias has an analytically zero gradient (softmax is invariant to a shift of
            # every logit of a row), so only its third may wander (bounded by Adam's step size below); a bug in the query / value bias path must still fail at 0.05
            d = v.numel() // 3
            for name, sl in (('q', slice(0, d)), ('v', slice(2 * d, 3 * d))):
                upd_ref = (ref_sd[k][sl] - sd[k][sl]).double()
                err = float((v.cpu().double()[sl] - ref_sd[k].double()[sl]).norm() / (upd_ref.norm() + 1e-12))
                assert err < 0.05, (k, name, err)
            # the key third: both trajectories are +-lr round-off walks (observed relative distance 2.2), so the only sound bound is
            # Adam's own: no element moves further than lr per step
            moved = float((v.cpu().double()[d:2 * d] - sd[k].double()[d:2 * d]).abs().max())
            assert moved <= 3 * 1e-3 * 1.01, (k, 'k', moved)
            continue
        upd_ref = (ref_sd[k] - sd[k]).double()
        err = float((v.cpu().double() - ref_sd[k].double()).norm() / (upd_ref.norm() + 1e-12))
        assert err < 0.05, (k, err)


@pytest.mark.parametrize('precision', ['bf16', 'fp32', 'fp32x3'])
def test_grad_norm_fused_step_counts_every_gradient_once(precision):
    """with the optimiser inside the backward (bf16: the matrix share of the norm comes from the weight-gradient
    epilogues) the reported global gradient norm (utils/misc.py:280-292) must equal the norm of the gradient arena — with a
    large contrastive weight the predictor's matrices carry a large share, so counting them twice shows at the 10 % level."""
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    cfg = R.vit_base_cfg(contrastive=True, **VITB)
    model = build(cfg, R.init_state_dict(cfg, seed=0), precision=precision).train()
    opt = FusedAdamW(model, lr=0.0, weight_decay=0.0, betas=(0.9, 0.95))
    model._ensure_engine(torch.device('cuda', 0))
    eng = opt.engine
    eng.set_loss_weights(0.01, 100.0, 1)      # (the contrastive term's gradient is small at initialisation: weight it up)
    B = 2
    v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=11)
    model.set_masking_noise(*R.masking_noise(B, cfg.num_patches, seed=12))
    runner = model._step_runner(B, 0.75, True, False, True)
    runner.load(v1.cuda(), v2.cuda())
    eng.optimizer_hparams(lr=0.0)
    runner.run()
    torch.cuda.synchronize()
    got = float(eng.losses[5])
    want = float(eng.grads.double().norm())
    pred = float(torch.cat([eng.g['predictor.0.weight'].flatten(), eng.g['predictor.3.weight'].flatten()]).double().norm())
    assert pred > 0.05 * want, (pred, want)          # the case is sensitive to the predictor's share
    assert abs(got - want) <= 2e-4 * want, (got, want, pred)


def _ref_adamw(params, lr, wd):
    groups = R.param_groups(params, wd)
    return torch.optim.AdamW([{'params': [params[n] for n in g['names']], 'weight_decay': g['weight_decay']}
                              for g in groups], lr=lr, betas=(0.9, 0.95))


def test_checkpoint_wire_format_both_ways(tmp_path):
    """SURVEY §8(f) row 2 (utils/misc.py:295-329): a checkpoint written the way the reference writes it
    ({'model','optimizer','epoch','scaler','args'}, torch.optim.AdamW state layout) resumes on the HIP path, and a
    checkpoint written by the HIP path resumes in a plain torch.optim.AdamW — the next step agrees either way."""
    from vit_ae_plus_plus_amd.utils import misc
    from vit_ae_plus_plus_amd.utils.train_one_epoch import train_one_stage_epoch
    cfg = R.RefConfig(contrastive=True, **MICRO)
    sd0 = R.init_state_dict(cfg, seed=3)
    lr, wd, B = 1e-3, 0.05, 2

    def batch(i):
        v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=300 + i)
        return v1, v2, R.masking_noise(B, cfg.num_patches, seed=400 + i)

    def hip_epoch(model, opt, i):
        v1, v2, noises = batch(i)
        model.set_masking_noise(*noises)
        args = argparse.Namespace(accum_iter=1, mask_ratio=0.75, contr_weight=0.001, lr=lr, min_lr=0.0, warmup_epochs=0,
                                  epochs=50, hip_graph=False, no_fused_step=False)
        return train_one_stage_epoch(model, [(v1, v2, torch.zeros(B))], opt, torch.device('cuda'), 0,
                                     misc.NativeScalerWithGradNormCount(), log_writer=None, args=args, edge_map_weight=0.01)

    # ---- reference-format checkpoint (CPU torch.optim.AdamW after one step) -> HIP path
    tr = T.RefTrainer(cfg, sd0, lr=lr, weight_decay=wd)
    v1, v2, (n1, n2) = batch(0)
    tr.step(v1, v2, n1, n2, lr=lr, edge_map_weight=0.01, contr_weight=0.001)
    ref_ck = tmp_path / 'checkpoint-0.pth'
    torch.save({'model': tr.state_dict(), 'optimizer': tr.optimizer.state_dict(), 'epoch': 0,
                'scaler': torch.cuda.amp.GradScaler(enabled=False).state_dict(), 'args': argparse.Namespace(lr=lr)}, ref_ck)
    model = build(cfg, sd0)
    named = dict(model.named_parameters())
    opt = _ref_adamw(named, lr, wd)
    misc.load_model(argparse.Namespace(resume=str(ref_ck)), model, opt, misc.NativeScalerWithGradNormCount())
    stats = hip_epoch(model, opt, 1)
    v1, v2, (n1, n2) = batch(1)
    terms, _, _ = tr.step(v1, v2, n1, n2, lr=lr, edge_map_weight=0.01, contr_weight=0.001)
    close(stats['loss'], terms['loss'], 2e-4, 1e-7)
    ref_sd = tr.state_dict()
    for k in ('decoder_pred.weight', 'blocks.0.mlp.fc1.weight', 'patch_embed.proj.weight', 'predictor.3.weight'):
        upd = (ref_sd[k] - sd0[k]).double().norm()
        assert float((model.state_dict()[k].cpu().double() - ref_sd[k].double()).norm() / upd) < 0.05, k

    # ---- HIP-path checkpoint -> file -> fresh HIP model AND plain torch AdamW on the oracle
    out = tmp_path / 'out'
    out.mkdir()
    misc.save_model(argparse.Namespace(output_dir=str(out)), 1, model, model, opt, misc.NativeScalerWithGradNormCount())
    ck = torch.load(out / 'checkpoint-1.pth', map_location='cpu', weights_only=False)
    assert set(ck) == {'model', 'optimizer', 'epoch', 'scaler', 'args'}
    assert set(ck['optimizer']) == {'state', 'param_groups'} and set(ck['optimizer']['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
    model2 = build(cfg, sd0)
    opt2 = _ref_adamw(dict(model2.named_parameters()), lr, wd)
    misc.load_model(argparse.Namespace(resume=str(out / 'checkpoint-1.pth')), model2, opt2, misc.NativeScalerWithGradNormCount())
    s_a, s_b = hip_epoch(model, opt, 2), hip_epoch(model2, opt2, 2)
    close(s_a['loss'], s_b['loss'], 1e-6, 1e-9)
    tr2 = T.RefTrainer(cfg, ck['model'], lr=lr, weight_decay=wd)
    tr2.optimizer.load_state_dict(ck['optimizer'])
    v1, v2, (n1, n2) = batch(2)
    terms2, _, _ = tr2.step(v1, v2, n1, n2, lr=lr, edge_map_weight=0.01, contr_weight=0.001)
    close(s_a['loss'], terms2['loss'], 2e-4, 1e-7)
    k = 'decoder_pred.weight'
    upd = (tr2.state_dict()[k] - ck['model'][k]).double().norm()
    assert float((model.state_dict()[k].cpu().double() - tr2.state_dict()[k].double()).norm() / upd) < 0.05


def _one_step_vs_oracle(cfg, B, precision, seed, loss_tol, gnorm_tol):
    """Forward + backward of one batch through the drop-in model vs the CPU oracle on identical weights, inputs and
    masking noise: loss scalars and per-parameter gradient norms."""
    from vit_ae_plus_plus_amd.utils.train_one_epoch import compute_contrastive_loss
    sd = R.init_state_dict(cfg, seed=seed)
    model = build(cfg, sd, precision=precision)
    model.train(True)
    v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=1234)
    noises = R.masking_noise(B, cfg.num_patches, seed=77)
    params = R.make_leaf_params(sd)
    if cfg.contrastive:
        loss, pred, mask, p1, p2, z1, z2 = R.contr_forward(params, v1, v2, noises[0], noises[1], cfg, 0.75, 0.01)
        total = loss[0] + R.contrastive_loss(p1, p2, z1, z2, 0.001)
    else:
        loss, pred, mask = R.mae_forward(params, v1, noises[0], cfg, 0.75, 0.01)
        total = loss[0]
    total.backward()
    if cfg.contrastive:
        model.set_masking_noise(*noises)
        out = model(view1=v1.cuda(), view2=v2.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
        gl = out[0]
        gtotal = gl[0] + compute_contrastive_loss(argparse.Namespace(contr_weight=0.001), None, *out[3:])
    else:
        model.set_masking_noise(noises[0])
        gl, gpred, gmask = model(v1.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
        gtotal = gl[0]
        assert float(gmask.sum()) == float(mask.sum())
    gtotal.backward()
    lerr = max(abs(float(a.detach()) - float(b.detach())) / abs(float(b.detach())) for a, b in zip(gl[:3], loss[:3]))
    for a, b in zip(gl[:3], loss[:3]):
        close(a, b, loss_tol, 1e-7)
    named = dict(model.named_parameters())
    worst = 0.0
    for k, p in params.items():
        if not p.requires_grad or p.grad is None:
            continue
        ref = float(p.grad.double().norm())
        got = float(named[k].grad.double().norm())
        if ref > 1e-6 * max(1.0, float(total.detach())):
            worst = max(worst, abs(got - ref) / ref)
    print(f'{type(model).__name__} {cfg.volume_size} {precision}: worst loss error {lerr:.2e}, worst gradient-norm error {worst:.2e}')
    assert worst < gnorm_tol, worst


def _one_step_vs_reference_pins(fixture, cfg, B, seeds, precision, loss_tol, gnorm_tol, pred_tol):
    """One forward + backward of the drop-in model against pins produced by the REFERENCE model (oracle/gen_golden.py
    _one_step_pins): the loss scalars, mask sums, prediction samples, per-parameter gradient norms."""
    from vit_ae_plus_plus_amd.utils.train_one_epoch import compute_contrastive_loss
    g = load_golden(fixture)
    model = build(cfg, R.init_state_dict(cfg, seed=seeds[0]), precision=precision)
    model.train(True)
    v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=seeds[1])
    n1, n2 = R.masking_noise(B, cfg.num_patches, seed=seeds[2])
    if cfg.contrastive:
        model.set_masking_noise(n1, n2)
        loss, pred, mask, p1, p2, z1, z2 = model(view1=v1.cuda(), view2=v2.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
        contr = compute_contrastive_loss(argparse.Namespace(contr_weight=0.001), None, p1, p2, z1, z2)
        close(contr, g['contr_loss'], 20 * loss_tol, 1e-8)
    else:
        model.set_masking_noise(n1)
        loss, pred, mask = model(v1.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
        contr = torch.zeros((), device='cuda')
    got, ref = torch.stack(loss).detach().double().cpu().numpy(), g['losses']
    errs = np.abs(got[:3] - ref[:3]) / np.abs(ref[:3])
    assert (errs <= loss_tol).all(), (got, ref, errs)
    assert torch.equal(mask.sum(1).cpu(), t(g['mask_sum']))
    ps = pred[:, ::37, ::(1021 if pred.shape[-1] > 4096 else 127)].detach().double().cpu().numpy()
    assert np.abs(ps - g['pred_slice']).max() <= pred_tol * np.abs(g['pred_slice']).max()      # max-norm, relative to the prediction's scale
    (loss[0] + contr).backward()
    named = dict(model.named_parameters())
    worst, worst_name = 0.0, None
    for k, refn in zip(list(g['grad_names']), g['grad_norms']):
        gotn = float(named[str(k)].grad.double().norm())
        if refn > 1e-6 * max(1.0, float(ref[0])) and abs(gotn - refn) / refn > worst:
            worst, worst_name = abs(gotn - refn) / refn, str(k)
    assert worst < gnorm_tol, (worst, worst_name)
    print(f'{fixture} {precision}: worst loss error {errs.max():.2e}, worst gradient-norm error {worst:.2e}')
    return float(errs.max()), worst


@pytest.mark.parametrize('precision,loss_tol,gnorm_tol,pred_tol', [('fp32', 1e-4, 2e-3, 1e-3), ('fp32x3', 1e-4, 2e-3, 1e-3), ('bf16', 1e-3, 6e-3, 3e-2)])   # bf16: 3x observed (2.8e-4 / 1.6e-3)
def test_vitb_patch8_vs_reference_pins(precision, loss_tol, gnorm_tol, pred_tol):
    """The reference's SHIPPED configuration (config.ini:33 patch_size = 8 -> read_configs.py:38 -> model_factory.py:12):
    contrastive ViT-B on 96^3 x 4ch is 1728 patches, 433 encoder tokens (head dim 64) and 1729 decoder tokens (head dim 32: beyond
    the one-launch attention backward's LDS budget, so the two-kernel backward runs), decoder_pred 512 -> 2048.  Pins from the
    reference model itself (tests/golden/vitb_p8.npz): fp32 mode at the north-star tolerance, bf16 at its round-off."""
    cfg = R.vit_base_cfg(volume_size=(96, 96, 96), patch_size=8, in_chans=4, contrastive=True)
    _one_step_vs_reference_pins('vitb_p8.npz', cfg, 1, (0, 1234, 4321), precision, loss_tol, gnorm_tol, pred_tol)


@pytest.mark.parametrize('precision,loss_tol,gnorm_tol,pred_tol', [('fp32', 1e-4, 2e-3, 1e-3), ('fp32x3', 1e-4, 2e-3, 1e-3), ('bf16', 1e-4, 6e-3, 3e-2)])   # bf16: 3x observed (2.8e-5 / 1.9e-3)
def test_config4_vs_reference_pins(precision, loss_tol, gnorm_tol, pred_tol):
    """BASELINE config 4 (mae_vit_large_patch16 on 128^3 x 4ch, B = 1) against the reference model's own numbers
    (tests/golden/vitl_128.npz) — round 2 checked this configuration against the live oracle only."""
    cfg = R.vit_large_cfg(volume_size=(128, 128, 128), patch_size=16, in_chans=4, contrastive=False)
    _one_step_vs_reference_pins('vitl_128.npz', cfg, 1, (2, 1234, 77), precision, loss_tol, gnorm_tol, pred_tol)


@pytest.mark.parametrize('precision,loss_tol,gnorm_tol', [('fp32', 1e-4, 2e-3), ('bf16', 1e-4, 6e-3)])     # bf16: 3x observed (2.8e-5 / 1.9e-3)
def test_config4_vit_large_128(precision, loss_tol, gnorm_tol):
    """BASELINE config 4: ViT-L/16 autoencoder, 128^3 x 4ch (129 encoder / 513 decoder tokens — the decoder exceeds the
    one-launch attention backward's LDS budget, so the two-kernel path runs), B = 1, against the oracle."""
    cfg = R.vit_large_cfg(volume_size=(128, 128, 128), patch_size=16, in_chans=4, contrastive=False)
    _one_step_vs_oracle(cfg, 1, precision, seed=2, loss_tol=loss_tol, gnorm_tol=gnorm_tol)


@pytest.mark.parametrize('precision,loss_tol,gnorm_tol', [('fp32', 1e-4, 2e-3), ('bf16', 6e-4, 7e-3)])     # bf16: 3x observed (2.0e-4 / 2.1e-3)
def test_config5_anisotropic_egd_shape(precision, loss_tol, gnorm_tol):
    """BASELINE config 5: EGD-shape 192 x 192 x 32 x 1ch volumes, ViT-B (non-cubic patch grid 12 x 12 x 2).  The reference
    cannot construct this model, so the pin is the oracle's non-cubic generalisation."""
    cfg = R.vit_base_cfg(volume_size=(192, 192, 32), patch_size=16, in_chans=1, contrastive=True)
    _one_step_vs_oracle(cfg, 2, precision, seed=4, loss_tol=loss_tol, gnorm_tol=gnorm_tol)


ACT16 = dict(volume_size=(16, 16, 16), patch_size=4, in_chans=4, embed_dim=64, depth=2, num_heads=2,
             decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2)   # every contraction length % 64 == 0


@pytest.mark.parametrize('name,dims,precision,tol', [('generic', MICRO, 'fp32', 2e-4), ('act16', ACT16, 'bf16', 2e-2)])
@pytest.mark.parametrize('use_graph', [False, True])
def test_gradient_accumulation_matches_oracle(name, dims, precision, tol, use_graph):
    """accum_iter = 2 (utils/train_one_epoch.py:44-74: loss / accum_iter, optimizer step every second iteration) over
    four iterations against the oracle loop; 'act16' exercises the bf16-operand GEMM path's accumulate flags."""
    from vit_ae_plus_plus_amd.utils import misc
    from vit_ae_plus_plus_amd.utils.train_one_epoch import train_one_stage_epoch
    cfg = R.RefConfig(contrastive=True, **dims)
    sd = R.init_state_dict(cfg, seed=9)
    lr, wd, B = 2e-3, 0.05, 2
    batches, noises = [], []
    for i in range(4):
        v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=500 + i)
        batches.append((v1, v2, torch.zeros(B)))
        noises.append(R.masking_noise(B, cfg.num_patches, seed=600 + i))
    tr = T.RefTrainer(cfg, sd, lr=lr, weight_decay=wd)
    ref = T.train_one_stage_epoch_ref(tr, batches, 0, lr=lr, min_lr=0.0, warmup_epochs=0, epochs=50, mask_ratio=0.75,
                                      contr_weight=0.001, edge_map_weight=0.01, accum_iter=2, noises=noises)
    model = build(cfg, sd, precision=precision)
    if name == 'act16':
        model._ensure_engine(torch.device('cuda', 0))
        assert model.engine.act16
    model.set_masking_noise(*[n for pair in noises for n in pair])
    named = dict(model.named_parameters())
    opt = _ref_adamw(named, lr, wd)
    args = argparse.Namespace(accum_iter=2, mask_ratio=0.75, contr_weight=0.001, lr=lr, min_lr=0.0, warmup_epochs=0, epochs=50,
                              hip_graph=use_graph, no_fused_step=False)
    stats = train_one_stage_epoch(model, batches, opt, torch.device('cuda'), 0, misc.NativeScalerWithGradNormCount(),
                                  log_writer=None, args=args, edge_map_weight=0.01)
    for k in ('loss', 'reconstruction_loss', 'edge_map_loss', 'lr'):
        close(stats[k], ref[k], tol, 1e-7)
    ref_sd = tr.state_dict()
    for k in ('decoder_pred.weight', 'blocks.0.mlp.fc1.weight', 'blocks.1.attn.qkv.weight', 'patch_embed.proj.weight',
              'decoder_blocks.0.attn.proj.weight', 'norm.weight'):
        upd = (ref_sd[k] - sd[k]).double().norm()
        err = float((model.state_dict()[k].cpu().double() - ref_sd[k].double()).norm() / upd)
        assert err < (0.05 if precision == 'fp32' else 0.25), (k, err)


@pytest.mark.parametrize('comm,use_graph,dec_chunks', [(None, False, 1), (None, True, 1), ('bf16', False, 1), ('bf16', True, 1),
                                                       (None, True, 2), ('bf16', False, 2)])
def test_data_parallel_machinery_on_a_world_of_one(comm, use_graph, dec_chunks, monkeypatch):
    """The N > 1 step (per-phase graphs, bucketed all-reduce between them, buckets stepped on the optimiser stream as
    their exchange lands, optional bf16 wire) on a single-rank RCCL group: an all-reduce over one rank is the identity,
    so two steps must land where the single-process step lands (bf16 wire: to bf16 round-off of the gradients)."""
    import torch.distributed as dist
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    created = False
    if not dist.is_initialized():
        import os
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        created = True
    try:
        cfg = R.RefConfig(contrastive=True, **dict(ACT16, decoder_depth=2 if dec_chunks == 2 else ACT16['decoder_depth']))
        sd = R.init_state_dict(cfg, seed=11)
        B, outs = 2, []
        for ddp_on in (False, True):
            model = build(cfg, sd, precision='bf16')
            opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05)
            model._ensure_engine(torch.device('cuda', 0))
            eng = opt.engine
            if ddp_on and dec_chunks == 2:
                # default bucket structure (encoder: small last bucket) + the decoder backward in two phases / buckets
                monkeypatch.setenv('VITAE_DEC_CHUNKS', '2')
                red = model.enable_data_parallel(torch.device('cuda', 0), force=True, comm_dtype=torch.bfloat16 if comm else None)
                assert eng.dec_chunks == 2 and eng.N_PHASES == 2 + eng.enc_chunks + 1 and len(red.ranges) == eng.N_PHASES
                flat = sorted(r for parts in red.ranges for r in parts)
                assert flat[0][0] == 0 and flat[-1][1] == eng.n_total and all(a[1] == b[0] for a, b in zip(flat, flat[1:]))
            elif ddp_on:
                red = model.enable_data_parallel(torch.device('cuda', 0), force=True,
                                                 comm_dtype=torch.bfloat16 if comm else None, enc_chunks=2)
                assert red is not None and len(red.ranges) == 4
            eng.set_loss_weights(0.01, 0.001, 1, 1)
            for step in range(2):
                v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=700 + step)
                model.set_masking_noise(*R.masking_noise(B, cfg.num_patches, seed=800 + step))
                runner = model._step_runner(B, 0.75, True, False, use_graph)
                runner.load(v1, v2)
                eng.optimizer_hparams(lr=1e-3)
                runner.run()
            torch.cuda.synchronize()
            outs.append(({k: v.detach().clone() for k, v in model.state_dict().items()}, eng.losses.cpu().tolist()))
        (a, la), (b, lb) = outs
        close(lb[0], la[0], 2e-3 if comm else 1e-5, 1e-7)
        close(lb[5], la[5], 2e-2 if comm else 1e-4, 1e-7)          # grad norm
        for k in ('decoder_pred.weight', 'blocks.0.mlp.fc1.weight', 'blocks.1.attn.qkv.weight', 'patch_embed.proj.weight',
                  'predictor.3.weight', 'cls_token', 'norm.weight'):
            upd = (a[k].double() - sd[k].double().cuda()).norm()
            err = float((a[k].double() - b[k].double()).norm() / upd)
            assert err < (0.1 if comm else 2e-3), (k, err)
    finally:
        if created:
            dist.destroy_process_group()


def test_grouped_weight_gradients_match_the_paired_launches():
    """engine.wgrad_group_min: with many token rows the four weight gradients of a block leave as ONE launch at its end
    (vitae_wgrad_group_bt) and the Linears' backward launches compute input gradients only — same arithmetic, another launch
    structure: one fused step must land on the same losses and gradients (up to the summation order of split reductions)."""
    cfg = R.vit_base_cfg(contrastive=True, **VITB)
    sd = R.init_state_dict(cfg, seed=0)
    v1, v2 = R.synthetic_views((2, 4, 96, 96, 96), seed=1234)
    n1, n2 = R.masking_noise(2, cfg.num_patches, seed=4321)
    res = []
    for rows in (1 << 30, 1):
        model = build(cfg, sd, precision='bf16').train()
        eng = model._ensure_engine(torch.device('cuda', 0))
        eng.wgrad_group_min = rows
        eng.side_min_rows = 1        # (round 6: grouped launches exist only where the branch streams fork — few rows run as one chain)
        model.set_masking_noise(n1, n2)
        loss, pred, mask, p1, p2, z1, z2 = model(view1=v1.cuda(), view2=v2.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
        loss[0].backward()
        torch.cuda.synchronize()
        assert eng._grouped(256, 768) == (rows == 1) and ('encdx_16b' in eng.buf) == (rows == 1)
        res.append(([float(x.detach()) for x in loss],
                    {k: p.grad.detach().double().clone() for k, p in model.named_parameters() if p.requires_grad}))
        del model
    close(res[0][0], res[1][0], 1e-6, 1e-9)
    worst = {}
    for k, a in res[0][1].items():
        worst[k] = float((a - res[1][1][k]).norm() / (a.norm() + 1e-30))
    # the same products in another launch structure: the input-gradient GEMMs choose their own k-split when they run alone, their
    # last-bit differences become bf16 roundings of the running gradient, and 20 blocks later the gradients agree to bf16
    # round-off (the same level as run-to-run differences of the paired path under another split)
    bad = {k: e for k, e in worst.items() if e > 5e-3}
    assert not bad, bad


def test_host_batches_are_double_buffered_and_match_device_batches():
    """Pinned host batches go through the copy stream into alternating input slots (one captured graph per slot);
    three graph-replayed steps must land exactly where the same steps fed from device tensors land."""
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    cfg = R.RefConfig(contrastive=True, **ACT16)
    sd = R.init_state_dict(cfg, seed=13)
    B, finals = 2, []
    for host in (False, True):
        model = build(cfg, sd, precision='bf16')
        opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05)
        model._ensure_engine(torch.device('cuda', 0))
        eng = opt.engine
        eng.set_loss_weights(0.01, 0.001, 1, 1)
        runner = model._step_runner(B, 0.75, True, False, True)
        slots = set()
        for step in range(3):
            v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=900 + step)
            model.set_masking_noise(*R.masking_noise(B, cfg.num_patches, seed=950 + step))
            if host:
                runner.load(v1.pin_memory(), v2.pin_memory())
            else:
                runner.load(v1.cuda(), v2.cuda())
            slots.add(runner.st['cur'])
            eng.optimizer_hparams(lr=1e-3)
            runner.run()
        torch.cuda.synchronize()
        assert slots == ({0, 1} if host else {0})
        finals.append((eng.losses.cpu().tolist(), model.state_dict()['blocks.0.mlp.fc1.weight'].clone()))
    close(finals[1][0][0], finals[0][0][0], 1e-5, 1e-7)
    # the two runs execute the same kernels on the same values; what may differ is the arrival order of the float atomics that
    # collect bias / LayerNorm-parameter gradients (last-bit differences after step 0, which AdamW's normalised update amplifies on
    # elements with a near-zero gradient): the weights agree to a small fraction of the distance they have moved
    w0 = sd['blocks.0.mlp.fc1.weight'].double().cuda()
    moved = float((finals[0][1].double() - w0).norm())
    assert float((finals[0][1].double() - finals[1][1].double()).norm()) < 2e-3 * moved


def test_recurring_device_batches_are_read_in_place():
    """A device batch whose tensors come back gets a graph captured on its own addresses (no staging copy) from the
    second time on; it must read the LIVE contents of those tensors and land exactly where staged batches land."""
    from vit_ae_plus_plus_amd.model.vit_autoenc import _DirectSlot
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    cfg = R.RefConfig(contrastive=True, **ACT16)
    sd = R.init_state_dict(cfg, seed=17)
    B, finals = 2, []
    for in_place in (False, True):
        model = build(cfg, sd, precision='bf16')
        opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05)
        model._ensure_engine(torch.device('cuda', 0))
        eng = opt.engine
        eng.set_loss_weights(0.01, 0.001, 1, 1)
        runner = model._step_runner(B, 0.75, True, False, True)
        shape = (B, cfg.in_chans, *cfg.volume_size)
        buf1, buf2 = torch.empty(shape, device='cuda'), torch.empty(shape, device='cuda')
        alive, direct = [], []
        for step in range(4):
            v1, v2 = R.synthetic_views(shape, seed=700 + step)
            model.set_masking_noise(*R.masking_noise(B, cfg.num_patches, seed=750 + step))
            if in_place:        # same tensors every step, new contents
                buf1.copy_(v1); buf2.copy_(v2)
                runner.load(buf1, buf2)
            else:               # fresh addresses every step (kept alive so that the allocator cannot hand them out again)
                alive.append((v1.cuda(), v2.cuda()))
                runner.load(*alive[-1])
            direct.append(isinstance(runner.slot, _DirectSlot))
            eng.optimizer_hparams(lr=1e-3)
            runner.run()
        torch.cuda.synchronize()
        assert direct == ([False, True, True, True] if in_place else [False] * 4)
        finals.append((eng.losses.cpu().tolist(), model.state_dict()['blocks.0.mlp.fc1.weight'].clone()))
    close(finals[1][0][0], finals[0][0][0], 1e-5, 1e-7)
    # four Adam steps at lr 1e-3: stale or misplaced input would move EVERY weight at the 1e-3 level.  The order of the float atomics
    # (token / bias gradients) usually leaves the two runs bit-identical and otherwise moves a typical weight by ~1e-6 — and a rare
    # one by far more: an element whose gradient is next to zero sits on a step of m / (sqrt(v) + eps), and with the moments stored in
    # bf16 (round 6) next to a rounding boundary as well (seen once in a full-suite run: max 5.6e-5).  So the sharp bound is on the
    # root mean square and the bound on the worst element is loose.
    diff = (finals[0][1] - finals[1][1]).float()
    stats = (float(diff.pow(2).mean().sqrt()), float(diff.abs().max()), int((diff.abs() > 1e-5).sum()))
    assert stats[0] < 5e-6 and stats[1] < 3e-4, stats


def test_small_steps_are_one_chain_and_forked_branches_give_the_same_step(monkeypatch):
    """Round 6: below ``side_min_rows`` decoder token rows the step's branches (target edge map, predictor, per-bucket optimiser,
    grouped weight gradients) are issued on the current stream — the captured graph is one chain on one hardware queue; with the
    branch streams forced on, the same three steps land on the same losses and weights."""
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    cfg = R.RefConfig(contrastive=True, **ACT16)
    sd = R.init_state_dict(cfg, seed=23)
    B, outs = 2, {}
    shape = (B, cfg.in_chans, *cfg.volume_size)
    for sel in ('auto', 'all', 'oside,side'):
        monkeypatch.setenv('VITAE_SIDE_STREAMS', sel)
        model = build(cfg, sd, precision='bf16')
        opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05)
        model._ensure_engine(torch.device('cuda', 0))
        eng = opt.engine
        eng.set_loss_weights(0.01, 0.001, 1, 1)
        runner = model._step_runner(B, 0.75, True, False, True)
        for step in range(3):
            v1, v2 = R.synthetic_views(shape, seed=900 + step)
            model.set_masking_noise(*R.masking_noise(B, cfg.num_patches, seed=950 + step))
            runner.load(v1.cuda(), v2.cuda())
            eng.optimizer_hparams(lr=1e-3)
            runner.run()
        torch.cuda.synchronize()
        assert eng.forked_branches() == {'auto': [], 'all': ['oside', 'pside', 'side', 'wside'], 'oside,side': ['oside', 'side']}[sel]
        outs[sel] = (eng.losses.cpu().tolist(), model.state_dict()['decoder_blocks.0.mlp.fc2.weight'].clone())
        if sel == 'auto':       # the rule itself: rows of the decoder's token matrix against the threshold; data parallel always forks
            assert eng.Md == B * (cfg.num_patches + 1) < eng.side_min_rows
            eng.side_min_rows = eng.Md
            assert eng.forked_branches() == ['oside', 'pside', 'side', 'wside']
            eng.side_min_rows = eng.Md + 1
            assert eng.forked_branches() == []
            eng._ddp_active = True
            assert eng.forked_branches() == ['oside', 'pside', 'side', 'wside']
            eng._ddp_active = False
    for sel in ('all', 'oside,side'):
        for k in range(6):
            close(outs[sel][0][k], outs['auto'][0][k], 1e-5, 1e-7)
        diff = (outs[sel][1] - outs['auto'][1]).float()
        assert float(diff.pow(2).mean().sqrt()) < 5e-6 and float(diff.abs().max()) < 3e-4


# --------------------------------------------------------------------------- the benchmarked path itself, pinned
def _b4_batch(cfg, it):
    v1, v2 = R.synthetic_views((4, 4, 96, 96, 96), seed=1234 + it)
    return v1, v2, R.masking_noise(4, cfg.num_patches, seed=4321 + it)


# Tolerances of precision='bf16' against the fp32 reference pins (tests/golden/vitb_b4.npz, produced by the reference model
# itself): operands of every dense contraction are rounded to 8 mantissa bits, accumulation / master weights / optimiser
# state stay fp32.  At the first step the prediction is ~0, so the losses are insensitive to the model (observed 4e-6 /
# 2e-5); after AdamW steps (lr 1e-4, the bench's) the prediction carries the rounding.  Gradient norms: 5e-3 per parameter.
B4_LOSS_RTOL_STEP0, B4_LOSS_RTOL_LATER, B4_EDGE_RTOL, B4_CONTR_RTOL, B4_GRAD_RTOL = 1e-4, 1e-4, 6e-4, 2e-2, 5e-3
# Round 5: with the decoder's fc1 on two-plane weights (engine._init_w2; the rounding of THAT weight carried 1.2e-4 of the 1.5e-4) the
# total loss holds the north star's 1e-4 on every step (observed <= 4.7e-5; raw edge loss <= 1.5e-4, was 4.5e-4-6.6e-4).
# (observed on MI355X, round 2: total loss 2e-5 at the first step and <= 1.6e-4 later, reconstruction loss <= 4e-6, raw edge loss
# <= 6.4e-4, contrastive term (magnitude 1e-5) <= 5.7e-3, gradient norms <= 1.2e-3; the fp32 mode: everything <= 7e-7)


def test_bf16_optimizer_moments_checkpoint_and_beta_guard():
    """Round 6: in precision='bf16' AdamW's moments are STORED in bf16 (engine.state16).  (1) The FusedAdamW state dict carries them in fp32
    and a fresh model that loads it takes the same next step bit for bit; (2) an optimiser whose betas are closer to 1 than 1 - 2^-6
    (torch's default beta2 = 0.999) keeps fp32 moments — adopted torch.optim.AdamW included — and fp32 modes never use bf16 storage."""
    from vit_ae_plus_plus_amd.optim import FusedAdamW, adopt
    cfg = R.RefConfig(contrastive=True, **ACT16)
    sd0 = R.init_state_dict(cfg, seed=3)
    B = 2

    def make(betas=(0.9, 0.95), precision='bf16'):
        model = build(cfg, sd0, precision=precision).train()
        opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05, betas=betas)
        model._ensure_engine(torch.device('cuda', 0))
        eng = opt.engine
        eng.set_loss_weights(0.01, 0.001, 1)
        return model, opt, eng

    def step(model, eng, i):
        v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=700 + i)
        model.set_masking_noise(*R.masking_noise(B, cfg.num_patches, seed=800 + i))
        runner = model._step_runner(B, 0.75, True, False, False)
        runner.load(v1.cuda(), v2.cuda())
        eng.optimizer_hparams(lr=1e-3)
        runner.run()
        torch.cuda.synchronize()

    model, opt, eng = make()
    assert eng.state16 and eng.opt_state['exp_avg'].dtype == torch.bfloat16 and eng.opt_state['exp_avg_sq'].dtype == torch.bfloat16
    for i in range(2):
        step(model, eng, i)
    sd_opt, sd_model = opt.state_dict(), {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert sd_opt['fused_adamw']['exp_avg'].dtype == torch.float32 and sd_opt['fused_adamw']['step'] == 2
    assert float(sd_opt['fused_adamw']['exp_avg'].abs().max()) > 0
    model2, opt2, eng2 = make()
    model2.load_state_dict(sd_model)
    opt2.load_state_dict(sd_opt)
    assert torch.equal(eng2.opt_state['exp_avg'], eng.opt_state['exp_avg']) and torch.equal(eng2.opt_state['exp_avg_sq'], eng.opt_state['exp_avg_sq'])
    before = eng.params.clone()
    step(model, eng, 2); step(model2, eng2, 2)
    upd = float((eng.params - before).norm())
    # (the same step up to the summation order of the atomically accumulated bias / token gradients)
    assert upd > 0 and float((eng.params - eng2.params).norm()) < 1e-3 * upd
    assert float((eng.opt_state['exp_avg'].float() - eng2.opt_state['exp_avg'].float()).norm()) < 1e-3 * float(eng.opt_state['exp_avg'].float().norm())
    # the guard
    _, _, eng3 = make(betas=(0.9, 0.999))
    assert not eng3.state16 and eng3.opt_state['exp_avg_sq'].dtype == torch.float32
    _, _, eng4 = make(precision='fp32')
    assert not eng4.state16
    model5 = build(cfg, sd0, precision='bf16').train()
    model5._ensure_engine(torch.device('cuda', 0))
    topt = _ref_adamw(dict(model5.named_parameters()), 1e-3, 0.05)          # betas (0.9, 0.95)
    assert adopt(topt, model5) is topt and model5.engine.state16
    eng.betas = (0.9, 0.999)
    with pytest.raises(Exception):
        eng.optimizer_hparams(lr=1e-3)                                       # a moment that slow would stall in bf16 storage: refused


def test_two_plane_weights_follow_the_bucketed_optimiser():
    """the two-plane forward reads BOTH planes of a weight from ``hilo`` (never the bf16 shadow), and the only thing that
    keeps them current is ``refresh_w2`` behind every AdamW bucket.  After fused steps with the optimiser inside the backward the hi
    plane must equal the shadow bit for bit and the lo plane bf16(W - hi); a range that cuts through a two-plane tensor is refused
    instead of silently leaving it stale."""
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    from vit_ae_plus_plus_amd._abi import VitaeError
    cfg = R.vit_base_cfg(contrastive=True, **VITB)
    model = build(cfg, R.init_state_dict(cfg, seed=0), precision='bf16').train()
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05, betas=(0.9, 0.95))
    model._ensure_engine(torch.device('cuda', 0))
    eng = opt.engine
    assert eng._w2, 'bf16 mode carries two-plane weights by default (decoder fc1)'
    eng.set_loss_weights(0.01, 0.001, 1)
    B = 2
    runner = model._step_runner(B, 0.75, True, False, True)
    for it in range(3):
        v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=50 + it)
        model.set_masking_noise(*R.masking_noise(B, cfg.num_patches, seed=60 + it))
        runner.load(v1.cuda(), v2.cuda())
        eng.optimizer_hparams(lr=1e-3)
        runner.run()
    torch.cuda.synchronize()
    for name, hilo in eng._w2.items():
        K = hilo.shape[1] // 2
        w, hi, lo = eng.p[name], hilo[:, :K], hilo[:, K:]
        assert torch.equal(hi, eng.p16[name]), name                              # the plane the forward multiplies by IS the shadow's value
        assert torch.equal(hi, w.to(torch.bfloat16)), name                       # ... and the shadow follows the master
        assert torch.equal(lo, (w - hi.float()).to(torch.bfloat16)), name
        assert float((w - R.init_state_dict(cfg, seed=0)[name].cuda()).abs().max()) > 0      # (the weights did move)
    off0, ln, stride, count = eng._w2_groups[0][:4]
    with pytest.raises(VitaeError):
        eng.refresh_w2(lo=off0 + 4, hi=off0 + stride * count)                    # cuts through the first tensor
    with pytest.raises(VitaeError):
        eng.refresh_w2(lo=0, hi=off0 + ln // 2)
    eng.refresh_w2(lo=off0, hi=off0 + ln)                                         # exactly one tensor: fine
    eng.refresh_w2(lo=0, hi=off0)                                                 # nothing touched: fine
    torch.cuda.synchronize()


# fp32x3 (split-operand bf16 MFMA) is held to the fp32 mode's bounds; 'all': the step's branches FORKED (what batch >= 16 and data parallel
# run; batch 4 itself is one chain since round 6) must hold the same pins
@pytest.mark.parametrize('precision,branches', [('bf16', 'auto'), ('fp32', 'auto'), ('fp32x3', 'auto'), ('bf16', 'all')])
def test_bench_workload_b4_fused_graph_vs_reference_pins(precision, branches, monkeypatch):
    """BASELINE config 2 at the batch the metric is quoted on (B = 4, contrastive ViT-B/16, 96^3 x 4ch) through the
    route bench.py times — the fused optimisation step replayed from a HIP graph: first-step loss scalars, per-parameter
    gradient norms and the loss trajectory of three AdamW steps against pins from the reference's own model (SURVEY §8c
    item 2; reference model/vit_autoenc.py:205-238, utils/train_one_epoch.py:52-75)."""
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    monkeypatch.setenv('VITAE_W2', 'dec.fc1')      # the bf16 bounds below were recorded with this class on two planes (the default): pinned
    monkeypatch.setenv('VITAE_SIDE_STREAMS', branches)
    g = load_golden('vitb_b4.npz')
    B, steps, lr, wd, mask_ratio, edge_w, contr_w = [float(v) for v in g['hp']]
    B, steps = int(B), int(steps)
    cfg = R.vit_base_cfg(contrastive=True, **VITB)
    model = build(cfg, R.init_state_dict(cfg, seed=0), precision=precision).train()
    opt = FusedAdamW(model, lr=lr, weight_decay=wd, betas=(0.9, 0.95))
    model._ensure_engine(torch.device('cuda', 0))
    eng = opt.engine
    eng.set_loss_weights(edge_w, contr_w, 1)
    bf = precision == 'bf16'
    # gradients of the first step: one graph-replayed step WITHOUT the optimiser
    v1, v2, (n1, n2) = _b4_batch(cfg, 0)
    model.set_masking_noise(n1, n2)
    r0 = model._step_runner(B, mask_ratio, False, False, True)
    r0.load(v1.cuda(), v2.cuda())
    r0.run()
    torch.cuda.synchronize()
    assert eng.forked_branches() == ([] if branches == 'auto' else ['oside', 'pside', 'side', 'wside'])
    assert torch.equal(eng.buf['mask'][:B].sum(1).cpu(), t(g['mask_sum']))
    named = dict(model._trainable_named)
    worst = 0.0
    for k, ref in zip(g['grad_names'], g['grad_norms']):
        got = float(eng.g[str(k)].double().norm())
        worst = max(worst, abs(got - ref) / (ref + 1e-12))
        assert abs(got - ref) <= (B4_GRAD_RTOL if bf else 2e-3) * ref + 1e-9, (str(k), got, ref)
    # the trajectory: three optimiser steps, then the losses of a fourth batch
    runner = model._step_runner(B, mask_ratio, True, False, True)
    errs = []
    for it in range(steps + 1):
        v1, v2, (n1, n2) = _b4_batch(cfg, it)
        model.set_masking_noise(n1, n2)
        runner.load(v1.cuda(), v2.cuda())
        eng.optimizer_hparams(lr=lr)
        runner.run()
        got = eng.losses.cpu().tolist()
        want = g['losses'][it]           # [loss, raw edge, recon, percep, contr]
        rt = (B4_LOSS_RTOL_STEP0 if it == 0 else B4_LOSS_RTOL_LATER) if bf else 1e-4
        for i in (0, 2):
            assert abs(got[i] - want[i]) <= rt * abs(want[i]), (it, i, got, want)
        assert abs(got[1] - want[1]) <= (B4_EDGE_RTOL if bf else 1e-4) * abs(want[1]) + 1e-9, (it, got, want)
        assert abs(got[4] - want[4]) <= (B4_CONTR_RTOL if bf else 1e-4) * abs(want[4]) + 1e-8, (it, got, want)
        errs.append([abs(got[i] - want[i]) / (abs(want[i]) + 1e-12) for i in (0, 1, 2, 4)])
    print(f'{precision}: worst grad-norm error {worst:.2e}; loss errors per step [loss, edge, recon, contr] {errs}')


def test_bf16_with_the_decoder_on_two_plane_weights_holds_every_loss_term(monkeypatch):
    """VITAE_W2='decoder,enc.proj' (round 6): every Linear of the decoder and the encoder's attention projection multiply by hi + lo
    planes of their weights in the forward.  tools/bf16_rounding_ablation.py: the raw edge term of the bf16 route is carried by the
    rounding of the decoder's weights as a whole (single classes partly cancel: with decoder_pred alone on two planes it gets WORSE),
    so only the whole decoder brings it to the north star's 1e-4 — measured total 1.5e-5, raw edge 5.4e-5, reconstruction 1e-6 over
    the reference's pinned four-step trajectory, at +3.4 % of the batch-4 step (tools/w2_parity.py), which is why it is an option and
    not the default."""
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    monkeypatch.setenv('VITAE_W2', 'decoder,enc.proj')
    g = load_golden('vitb_b4.npz')
    B, steps, lr, wd, mask_ratio, edge_w, contr_w = [float(v) for v in g['hp']]
    B, steps = int(B), int(steps)
    cfg = R.vit_base_cfg(contrastive=True, **VITB)
    model = build(cfg, R.init_state_dict(cfg, seed=0), precision='bf16').train()
    opt = FusedAdamW(model, lr=lr, weight_decay=wd, betas=(0.9, 0.95))
    model._ensure_engine(torch.device('cuda', 0))
    eng = opt.engine
    assert len(eng._w2) == 4 * cfg.decoder_depth + 2 + cfg.depth
    eng.set_loss_weights(edge_w, contr_w, 1)
    runner = model._step_runner(B, mask_ratio, True, False, True)
    worst = [0.0, 0.0, 0.0]
    for it in range(steps + 1):
        v1, v2, (n1, n2) = _b4_batch(cfg, it)
        model.set_masking_noise(n1, n2)
        runner.load(v1.cuda(), v2.cuda())
        eng.optimizer_hparams(lr=lr)
        runner.run()
        got, want = eng.losses.cpu().tolist(), g['losses'][it]
        for j, i in enumerate((0, 1, 2)):
            worst[j] = max(worst[j], abs(got[i] - want[i]) / abs(want[i]))
    print(f'two-plane decoder: worst rel err total {worst[0]:.2e} raw edge {worst[1]:.2e} recon {worst[2]:.2e}')
    assert worst[0] <= 5e-5 and worst[1] <= 1e-4 and worst[2] <= 1e-5, worst


def test_two_batch_sizes_on_one_engine_keep_their_graphs_valid():
    """a captured graph holds raw workspace addresses; a step at another batch size (smaller last batch, an eval
    call) must not leave the first runner replaying into freed memory.  Workspaces are kept per (batch, keep) and evicting
    one drops the graphs."""
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    cfg = R.RefConfig(contrastive=True, **MICRO)
    sd = R.init_state_dict(cfg, seed=5)

    def run(seq, ws_max=None):
        model = build(cfg, sd)
        opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05)
        model._ensure_engine(torch.device('cuda', 0))
        eng = opt.engine
        if ws_max is not None:
            eng._WS_MAX = ws_max
        eng.set_loss_weights(0.01, 0.001, 1)
        out = []
        for i, B in enumerate(seq):
            v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=100 + i)
            n1, n2 = R.masking_noise(B, cfg.num_patches, seed=200 + i)
            model.set_masking_noise(n1, n2)
            runner = model._step_runner(B, 0.75, True, False, True)
            runner.load(v1, v2)
            eng.optimizer_hparams(lr=1e-3)
            runner.run()
            # churn the caching allocator so that a stale graph would scribble over something visible
            junk = [torch.randn(1 << 18, device='cuda') for _ in range(4)]
            del junk
            out.append(eng.losses.cpu().tolist()[:6])
        return out, eng

    seq = [2, 2, 3, 2, 1, 3, 2]
    want, _ = run(seq)                      # workspaces cached: graphs stay valid
    got, eng = run(seq, ws_max=1)           # every switch evicts: graphs dropped and recaptured
    assert eng.ws_gen > 0
    for a, b in zip(want, got):
        close(a, b, 1e-6, 1e-9)
    # eager reference of the same sequence
    model = build(cfg, sd)
    tr = T.RefTrainer(cfg, sd, lr=1e-3, weight_decay=0.05)
    for i, B in enumerate(seq):
        v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=100 + i)
        n1, n2 = R.masking_noise(B, cfg.num_patches, seed=200 + i)
        terms, norm, _ = tr.step(v1, v2, n1, n2, lr=1e-3, mask_ratio=0.75, edge_map_weight=0.01, contr_weight=0.001)
        close(want[i][0] + want[i][4], terms['loss'], 5e-4, 1e-7)


@pytest.mark.parametrize('comm', [False, True])
@pytest.mark.parametrize('use_graph', [False, True])
def test_native_rccl_exchange_inside_the_step_graph(comm, use_graph):
    """the C-ABI exchange (vitae_ddp_init / _allreduce_bucket / _wait: RCCL called directly on a side HIP
    stream) on a communicator of ONE rank.  With the graph route the collectives are nodes of the single captured step graph.
    An all-reduce over one rank is the identity: two steps must land where the single-process step lands."""
    from vit_ae_plus_plus_amd._abi import lib
    from vit_ae_plus_plus_amd.optim import FusedAdamW
    assert lib.vitae_ddp_available()
    cfg = R.RefConfig(contrastive=True, **ACT16)
    sd = R.init_state_dict(cfg, seed=11)
    B, outs = 2, []
    for ddp_on in (False, True):
        model = build(cfg, sd, precision='bf16')
        opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05)
        model._ensure_engine(torch.device('cuda', 0))
        eng = opt.engine
        if ddp_on:
            red = model.enable_data_parallel(torch.device('cuda', 0), force=True, comm_dtype=torch.bfloat16 if comm else None,
                                             enc_chunks=2, native=True)
            assert red is not None and red.native and len(red.ranges) == 4 and lib.vitae_ddp_world_size() == 1
        eng.set_loss_weights(0.01, 0.001, 1, 1)
        for step in range(3):
            v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=700 + step)
            model.set_masking_noise(*R.masking_noise(B, cfg.num_patches, seed=800 + step))
            runner = model._step_runner(B, 0.75, True, False, use_graph)
            runner.load(v1, v2)
            eng.optimizer_hparams(lr=1e-3)
            runner.run()
        torch.cuda.synchronize()
        if ddp_on and use_graph:
            assert all(len(g) == 1 for g in runner.graphs.values())      # ONE graph per step, collectives inside
        outs.append(({k: v.detach().clone() for k, v in model.state_dict().items()}, eng.losses.cpu().tolist()))
        if ddp_on:
            del runner
            model.disable_data_parallel()      # graphs first, then the communicator they were captured with
            assert lib.vitae_ddp_world_size() == 0
    (a, la), (b, lb) = outs
    close(lb[0], la[0], 2e-3 if comm else 1e-5, 1e-7)
    close(lb[5], la[5], 2e-2 if comm else 1e-4, 1e-7)          # grad norm
    for k in ('decoder_pred.weight', 'blocks.0.mlp.fc1.weight', 'blocks.1.attn.qkv.weight', 'patch_embed.proj.weight',
              'predictor.3.weight', 'cls_token', 'norm.weight'):
        upd = (a[k].double() - sd[k].double().cuda()).norm()
        err = float((a[k].double() - b[k].double()).norm() / upd)
        assert err < (0.1 if comm else 2e-3), (k, err)


def test_native_exchange_refuses_more_fork_points_than_its_event_ring():
    """csrc/ddp.hip keeps 64 fork / join events; a captured step that needs more must be refused loudly (VITAE_ERR_UNSUPPORTED_SHAPE),
    not served by silently recycling an event that an earlier node of the same capture still depends on.  Eager launches may wrap."""
    import ctypes
    from vit_ae_plus_plus_amd._abi import lib, VitaeError
    assert lib.vitae_ddp_available()
    uid = ctypes.create_string_buffer(128)
    lib.vitae_ddp_unique_id(uid)
    lib.vitae_ddp_init(uid, 1, 0)
    try:
        buf = torch.ones(1024, device='cuda')
        main, comm = torch.cuda.Stream(), torch.cuda.Stream()
        for _ in range(70):                                   # eager: the ring wraps, every call is served
            lib.vitae_ddp_allreduce_bucket(buf.data_ptr(), buf.numel(), 0, main.cuda_stream, comm.cuda_stream)
        lib.vitae_ddp_wait(main.cuda_stream, comm.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(buf, torch.ones_like(buf))         # SUM over one rank
        g = torch.cuda.CUDAGraph()
        served, refused = 0, None
        with torch.cuda.graph(g, stream=main, capture_error_mode='thread_local'):
            for i in range(80):
                try:
                    lib.vitae_ddp_allreduce_bucket(buf.data_ptr(), buf.numel(), 0, main.cuda_stream, comm.cuda_stream)
                    served += 1
                except VitaeError as e:
                    refused = str(e)
                    break
            main.wait_stream(comm)                            # join the side stream so that the capture can end
        assert served == 64 and refused is not None and 'VITAE_ERR_UNSUPPORTED_SHAPE' in refused, (served, refused)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(buf, torch.ones_like(buf))
    finally:
        lib.vitae_ddp_destroy()

