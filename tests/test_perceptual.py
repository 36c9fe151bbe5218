"""the perceptual-loss hook (reference model/model_utils/perceptual_loss.py:46-77).  Parity with the
reference's weights (torchvision VGG16 + ckp-399.pth) is UNPINNED — neither exists here; the arithmetic is checked on random
weights against the plain-PyTorch restatement (oracle/percep_ref.py)."""
import pytest
import torch

from oracle import percep_ref as P


def _rand_sd(seed=0):
    from vit_ae_plus_plus_amd.model.model_utils.perceptual_loss import vgg_perceptual_loss
    torch.manual_seed(seed)
    m = vgg_perceptual_loss()
    return m, {k: v.detach().clone() for k, v in m.state_dict().items()}


def test_state_dict_layout_is_the_references():
    m, sd = _rand_sd()
    want = []
    for s, idxs in ((1, (0, 2)), (2, (5, 7)), (3, (10, 12, 14)), (4, (17, 19, 21))):
        for i in idxs:
            want += [f'slice{s}.{i}.weight', f'slice{s}.{i}.bias']
    assert list(sd.keys()) == want
    assert sd['slice1.0.weight'].shape == (64, 3, 3, 3) and sd['slice4.21.weight'].shape == (512, 512, 3, 3)
    assert not any(p.requires_grad for p in m.parameters())


def test_oracle_slices_and_shapes():
    _, sd = _rand_sd()
    x = torch.randn(2, 1, 3, 16, 24)
    outs = P.forward_one_view(sd, x)
    assert [tuple(o.shape) for o in outs] == [(6, 64, 16, 24), (6, 128, 8, 12), (6, 256, 4, 6), (6, 512, 2, 3)]
    assert float(P.perceptual_loss(sd, x.repeat(1, 2, 1, 1, 1), x.repeat(1, 2, 1, 1, 1))) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 2, 3, 16, 24), (1, 4, 8, 32, 32)])
def test_hip_perceptual_loss_vs_oracle(shape):
    m, sd = _rand_sd(1)
    # scale the random weights up so that deep features are not vanishingly small
    for k in sd:
        if k.endswith('weight'):
            sd[k] = sd[k] * 3.0
    m.load_state_dict(sd)
    m = m.cuda()
    g = torch.Generator().manual_seed(2)
    x1, x2 = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g)
    want = float(P.perceptual_loss(sd, x1, x2))
    m.chunk_bytes = 1 << 22                     # several chunks per channel
    got = float(m(x1.cuda(), x2.cuda()))
    assert abs(got - want) <= 3e-2 * abs(want), (got, want)      # bf16 operands through ten layers
    m.chunk_bytes = 2 << 30
    got2 = float(m(x1.cuda(), x2.cuda()))
    assert abs(got2 - got) <= 1e-5 * abs(got)
    assert float(m(x1.cuda(), x1.cuda())) == 0.0


@pytest.mark.gpu
def test_mae_forward_with_a_perceptual_weight():
    """model/vit_autoenc.py:228-231: losses[3] = w * percep(pred_vol, target_vol), added to losses[0]; no gradient."""
    import argparse
    from functools import partial
    from oracle import mae_ref as R
    from oracle.gen_golden import MICRO
    from vit_ae_plus_plus_amd.model import vit_autoenc as VA
    cfg = R.RefConfig(contrastive=False, **MICRO)
    sd = R.init_state_dict(cfg, seed=3)
    w = 0.5
    args = argparse.Namespace(use_imagenet=False, perceptual_weight=w)
    torch.manual_seed(4)
    m = VA.MaskedAutoencoderViT(volume_size=cfg.volume_size[0], patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim,
                                depth=cfg.depth, num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim,
                                decoder_depth=cfg.decoder_depth, decoder_num_heads=cfg.decoder_num_heads,
                                norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps), args=args)
    keys = list(m.state_dict().keys())
    assert 'perceptual_loss.slice1.0.weight' in keys and 'perceptual_loss.slice4.21.bias' in keys
    m.load_state_dict(sd, strict=False)
    vgg = {k[len('perceptual_loss.'):]: v.detach().clone() for k, v in m.state_dict().items() if k.startswith('perceptual_loss.')}
    m = m.cuda().train()
    v1, _ = R.synthetic_views((2, cfg.in_chans, *cfg.volume_size), seed=5)
    n1, _ = R.masking_noise(2, cfg.num_patches, seed=6)
    m.set_masking_noise(n1)
    loss, pred, mask = m(v1.cuda(), mask_ratio=0.75, edge_map_weight=0.01)
    want = w * float(P.perceptual_loss(vgg, m.unpatchify(pred.detach().cpu()), v1))
    assert abs(float(loss[3]) - want) <= 3e-2 * abs(want) + 1e-9, (float(loss[3]), want)
    assert abs(float(loss[0]) - (0.01 * float(loss[1]) + float(loss[2]) + float(loss[3]))) <= 1e-5 * abs(float(loss[0]))
    loss[0].backward()        # the term contributes no gradient and does not break the backward
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)


def test_random_vgg_weights_are_announced():
    """with no VGG16 weights loaded (the reference reads model/ckp-399.pth, absent here) the term is computed with
    random convolutions — the module says so once, and stops saying it after a state dict supplied slice*.* (CPU: no launch)."""
    import warnings
    from vit_ae_plus_plus_amd.model.model_utils.perceptual_loss import vgg_perceptual_loss
    m = vgg_perceptual_loss()
    assert not m._weights_loaded
    with pytest.warns(RuntimeWarning, match='RANDOM convolution weights'):
        m._warn_if_random()
    m2 = vgg_perceptual_loss()
    m2.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    assert m2._weights_loaded
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        m2._warn_if_random()
    holder = torch.nn.Module()
    holder.perceptual_loss = vgg_perceptual_loss()
    holder.load_state_dict({'perceptual_loss.' + k: v.clone() for k, v in m.state_dict().items()})
    assert holder.perceptual_loss._weights_loaded
