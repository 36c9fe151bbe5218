"""CPU: pin the oracle (oracle/mae_ref.py, oracle/train_ref.py) against fixtures produced by the
imported reference (oracle/gen_golden.py).  Tolerances are fp32 round-off (the oracle and the
reference run the same ATen CPU kernels in possibly different association orders)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import mae_ref as R
from oracle import train_ref as T
from oracle.gen_golden import MICRO, TINY, VITB


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# --------------------------------------------------------------------------- KATs
def test_pos_embed_kat():
    g = load_golden('kats.npz')
    mine = R.sincos_pos_embed_3d(12, (2, 2, 2), True)
    assert mine.dtype == np.float64
    np.testing.assert_array_equal(mine, g['pos_embed_12_2'])
    np.testing.assert_array_equal(R.sincos_pos_embed_3d(768, (6, 6, 6), True).astype(np.float32), g['pos_embed_768_6'])
    np.testing.assert_array_equal(R.sincos_pos_embed_3d(512, (6, 6, 6), True).astype(np.float32), g['pos_embed_512_6'])
    # SURVEY A.1 closed form: token 1 = (l,h,w)=(0,0,1) -> only the w block (cols 8-11) is non-trivial
    s, c, s2, c2 = np.sin(1.0), np.cos(1.0), np.sin(0.01), np.cos(0.01)
    close(mine[2], [0, 0, 1, 1, 0, 0, 1, 1, s, s2, c, c2], 1e-12, 1e-12)   # row 1 is cls offset
    close(mine[3], [s, s2, c, c2, 0, 0, 1, 1, 0, 0, 1, 1], 1e-12, 1e-12)   # token (0,1,0) -> h block first
    close(mine[5], [0, 0, 1, 1, s, s2, c, c2, 0, 0, 1, 1], 1e-12, 1e-12)   # token (1,0,0) -> l block second


def test_gaussian_taps_kat():
    g = load_golden('kats.npz')
    np.testing.assert_array_equal(R.gaussian_taps(2).numpy(), g['gauss_taps'])
    close(R.gaussian_taps(2).numpy()[:3], [0.00266126, 0.01344761, 0.04740850], 1e-5)


def test_patchify_kat():
    g = load_golden('kats.npz')
    v = t(g['patchify_in'])
    p = R.patchify(v, 2)
    np.testing.assert_array_equal(p.numpy(), g['patchify_out'])
    assert p[0, 0].tolist() == [0, 64, 1, 65, 4, 68, 5, 69, 16, 80, 17, 81, 20, 84, 21, 85]
    np.testing.assert_array_equal(R.unpatchify(p, 2, (2, 2, 2)).numpy(), g['patchify_in'])
    np.testing.assert_array_equal(g['unpatchify_roundtrip'], g['patchify_in'])
    # conv token order == patchify token order (l-major, then h, then w)
    sd = {'patch_embed.proj.weight': torch.zeros(12, 2, 2, 2, 2), 'patch_embed.proj.bias': torch.zeros(12)}
    sd['patch_embed.proj.weight'][0, 0, 0, 0, 0] = 1
    np.testing.assert_array_equal(R.patch_embed(v, sd, 2)[0, :, 0].numpy(), g['conv_token_order'])
    assert g['conv_token_order'].tolist() == [0, 2, 8, 10, 32, 34, 40, 42]


def test_patchify_noncubic_roundtrip():
    v = torch.randn(2, 3, 8, 4, 12)
    p = R.patchify(v, 4)
    assert p.shape == (2, 2 * 1 * 3, 64 * 3)
    assert torch.equal(R.unpatchify(p, 4, (2, 1, 3)), v)


def test_sobel_and_blur_kat():
    g = load_golden('kats.npz')
    np.testing.assert_array_equal(R.sobel_kernels().numpy(), g['sobel_weight'])
    x = t(g['sobel_in'])
    close(R.sobel_magnitude(x).numpy(), g['sobel_mag'], 1e-6)
    comp = g['sobel_components'][0, :, 1, 1, 1]
    assert comp.tolist() == [-32.0, 96.0, 288.0]
    close(g['sobel_mag'][0, 1, 1, 1], 305.26056, 1e-6)
    xr = t(g['sobel_rand_in'])
    close(R.sobel_magnitude(xr).numpy(), g['sobel_rand_mag'], 1e-5, 1e-5)
    close(R.gaussian_blur3d(xr).numpy(), g['blur_rand'], 1e-5, 1e-6)


def test_sobel_backward_nan_at_zero():
    """sqrt at exactly-zero gradient magnitude back-propagates NaN."""
    x = torch.zeros(1, 1, 4, 4, 4, requires_grad=True)
    R.sobel_magnitude(x).sum().backward()
    assert torch.isnan(x.grad).any()


# --------------------------------------------------------------------------- micro model
def _micro(name, contrastive):
    g = load_golden(name)
    cfg = R.RefConfig(contrastive=contrastive, **MICRO)
    sd = {k[3:]: t(g[k]) for k in g.files if k.startswith('sd/')}
    assert list(sd.keys()) == list(R.state_dict_spec(cfg).keys())
    for k, shp in R.state_dict_spec(cfg).items():
        assert tuple(sd[k].shape) == shp, k
    return g, cfg, sd


@pytest.mark.parametrize('name,contrastive', [('micro.npz', True), ('micro_mae.npz', False)])
def test_micro_forward_backward(name, contrastive):
    g, cfg, sd = _micro(name, contrastive)
    params = R.make_leaf_params(sd)
    mask_ratio, edge_w, contr_w = [float(v) for v in g['hp']]
    v1, v2, n1, n2 = (t(g[k]) for k in ('view1', 'view2', 'noise1', 'noise2'))
    trace = {}
    if contrastive:
        loss, pred, mask, p1, p2, z1, z2 = R.contr_forward(params, v1, v2, n1, n2, cfg, mask_ratio, edge_w,
                                                           trace=trace)
        contr = R.contrastive_loss(p1, p2, z1, z2, contr_w)
    else:
        loss, pred, mask = R.mae_forward(params, v1, n1, cfg, mask_ratio, edge_w, trace=trace)
        contr = torch.zeros(())
    (loss[0] + contr).backward()
    np.testing.assert_array_equal(mask.numpy(), g['mask'])
    close([float(l) for l in loss], g['losses'], 2e-6, 1e-7)
    close(float(contr), g['contr_loss'], 1e-5, 1e-9)
    close(pred.detach().numpy(), g['pred'], 1e-4, 2e-6)
    for k in ('patch_embed', 'latent', 'decoder_in', 'target', 'blurred', 'edge_target', 'edge_pred',
              'enc_block0', 'enc_block1', 'dec_block0', 'dec_block1'):
        close(trace[k].detach().numpy(), g['t/' + k], 1e-4, 5e-6)
    if contrastive:
        close(p1.detach().numpy(), g['p1'], 1e-4, 5e-6)
        close(p2.detach().numpy(), g['p2'], 1e-4, 5e-6)
        close(trace['latent2'].detach().numpy(), g['t2/latent'], 1e-4, 5e-6)
        close(trace['bn_state']['running_mean'].numpy(), g['bn_running_mean'], 1e-5, 1e-7)
        close(trace['bn_state']['running_var'].numpy(), g['bn_running_var'], 1e-5, 1e-7)
        assert int(trace['bn_state']['num_batches_tracked']) == int(g['bn_num_batches']) == 2
    n = 0
    for k, p in params.items():
        if not p.requires_grad:
            continue
        ref = g['grad/' + k]
        scale = max(1e-12, float(np.abs(ref).max()))
        close(p.grad.numpy() / scale, ref / scale, 1e-3, 2e-5)
        n += 1
    assert n == sum(1 for k in g.files if k.startswith('grad/'))


def test_micro_fp64_oracle_is_tighter_yardstick():
    g, cfg, sd = _micro('micro.npz', True)
    params = R.make_leaf_params(sd, dtype=torch.float64)
    mask_ratio, edge_w, contr_w = [float(v) for v in g['hp']]
    v1, v2, n1, n2 = (t(g[k]) for k in ('view1', 'view2', 'noise1', 'noise2'))
    loss, pred, *_ = R.contr_forward(params, v1.double(), v2.double(), n1, n2, cfg, mask_ratio, edge_w)
    close([float(l) for l in loss], g['losses'], 5e-6, 1e-7)


# --------------------------------------------------------------------------- config 1 epoch
def test_tiny_epoch_matches_reference_loop():
    g = load_golden('tiny_epoch.npz')
    lr, wd, edge_w, contr_w, mask_ratio, warm, epochs = [float(v) for v in g['hp']]
    cfg = R.RefConfig(contrastive=True, **TINY)
    tr = T.RefTrainer(cfg, R.init_state_dict(cfg, seed=0), lr=lr, weight_decay=wd)
    B, iters = 4, 2
    batches, noises = [], []
    for it in range(iters):
        v1, v2 = R.synthetic_views((B, 1, 64, 64, 64), seed=1234 + it)
        batches.append((v1, v2, None))
        noises.append(R.masking_noise(B, cfg.num_patches, seed=4321 + it))
    stats = T.train_one_stage_epoch_ref(tr, batches, int(g['epoch']), lr=lr, warmup_epochs=warm, epochs=epochs,
                                        mask_ratio=mask_ratio, contr_weight=contr_w, edge_map_weight=edge_w,
                                        noises=noises)
    assert set(stats) == {'lr', 'edge_map_loss', 'reconstruction_loss', 'perceptual_loss', 'contr_loss', 'loss'}
    for k, v in stats.items():
        close(v, g['stat/' + k], 2e-5, 1e-8)
    fin = tr.state_dict()
    for k in g.files:
        if k.startswith('norm/'):
            close(float(fin[k[5:]].double().norm()), g[k], 1e-5, 1e-7)
    close(fin['cls_token'].numpy(), g['final/cls_token'], 1e-4, 1e-6)
    close(fin['decoder_pred.bias'].numpy(), g['final/decoder_pred.bias'], 1e-3, 1e-6)
    close(fin['predictor.1.running_var'].numpy(), g['final/predictor.1.running_var'], 1e-4, 1e-7)


def test_lr_schedule():
    assert T.lr_at(0.0, 1e-3, 0.0, 40, 50) == 0.0
    close(T.lr_at(3.5, 1e-3, 0.0, 40, 50), 1e-3 * 3.5 / 40, 1e-15)
    close(T.lr_at(45.0, 1e-3, 1e-5, 40, 50), 1e-5 + (1e-3 - 1e-5) * 0.5, 1e-12)
    close(T.lr_at(50.0, 1e-3, 0.0, 40, 50), 0.0, 0, 1e-18)


# --------------------------------------------------------------------------- ViT-B pins (config 2 shape)
@pytest.mark.parametrize('tag', ['mae'])
def test_vitb_forward_pins(tag):
    """Forward-only (a few seconds on 8 cores); the gradient pins are exercised on the GPU box."""
    g = load_golden('vitb.npz')
    contrastive = tag == 'contr'
    cfg = R.vit_base_cfg(contrastive=contrastive, **VITB)
    sd = R.init_state_dict(cfg, seed=0)
    v1, v2 = R.synthetic_views((2, 4, 96, 96, 96), seed=1234)
    n1, n2 = R.masking_noise(2, cfg.num_patches, seed=4321)
    with torch.no_grad():
        loss, pred, mask = R.mae_forward(sd, v1, n1, cfg, 0.75, 0.01)
    close([float(l) for l in loss], g[f'{tag}/losses'], 1e-5, 1e-7)
    np.testing.assert_array_equal(mask.sum(1).numpy(), g[f'{tag}/mask_sum'])
    close(pred[:, ::37, ::1021].numpy(), g[f'{tag}/pred_slice'], 1e-3, 1e-5)


# --------------------------------------------------------------------------- the reference's shipped shape (patch 8) and config 4
@pytest.mark.parametrize('fixture,cfgf,B,seeds', [
    ('vitb_p8.npz', lambda: R.vit_base_cfg(volume_size=(96, 96, 96), patch_size=8, in_chans=4, contrastive=True), 1, (0, 1234, 4321)),
    ('vitl_128.npz', lambda: R.vit_large_cfg(volume_size=(128, 128, 128), patch_size=16, in_chans=4, contrastive=False), 1, (2, 1234, 77))])
def test_oracle_forward_at_p8_and_vitl_pins(fixture, cfgf, B, seeds):
    """Forward of the oracle against the pins taken from the reference's own model at config.ini's patch_size = 8 (L = 1728,
    433 / 1729 tokens) and at BASELINE config 4 (ViT-L/16, 128^3): loss scalars, mask sums, prediction samples."""
    g = load_golden(fixture)
    cfg = cfgf()
    sd = R.init_state_dict(cfg, seed=seeds[0])
    v1, v2 = R.synthetic_views((B, cfg.in_chans, *cfg.volume_size), seed=seeds[1])
    n1, n2 = R.masking_noise(B, cfg.num_patches, seed=seeds[2])
    with torch.no_grad():
        if cfg.contrastive:
            loss, pred, mask, p1, p2, z1, z2 = R.contr_forward(sd, v1, v2, n1, n2, cfg, 0.75, 0.01)
            close(float(R.contrastive_loss(p1, p2, z1, z2, 0.001)), g['contr_loss'], 1e-3, 1e-9)
            close(p1[::11, ::97].numpy(), g['p1_slice'], 2e-3, 1e-4)
        else:
            loss, pred, mask = R.mae_forward(sd, v1, n1, cfg, 0.75, 0.01)
    close([float(l) for l in loss], g['losses'], 2e-5, 1e-7)
    np.testing.assert_array_equal(mask.sum(1).numpy(), g['mask_sum'])
    close(pred[:, ::37, ::(1021 if pred.shape[-1] > 4096 else 127)].numpy(), g['pred_slice'], 1e-3, 1e-5)
