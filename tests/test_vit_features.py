"""SURVEY §8(f) row 1 — encoder-only feature extraction (reference model/vit.py:265-297,
utils/feature_extraction.py:9-45, post_training_utils/extract_ssl_features.py:111-135).

CPU part: the oracle restatement and the module's state-dict / checkpoint hand-off against the fixture taken from the
reference's own VisionTransformer3D.  GPU part: the HIP path against the same fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import mae_ref as R
from oracle import vit_ref as V
from oracle.gen_golden import MICRO, VITB

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'vit_features.npz')
ENC = {k: MICRO[k] for k in ('volume_size', 'patch_size', 'in_chans', 'embed_dim', 'depth', 'num_heads')}


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


def _module(cfg, precision='fp32'):
    from vit_ae_plus_plus_amd.model.vit import VisionTransformer3D
    vol = cfg.volume_size[0]
    return VisionTransformer3D(volume_size=vol, patch_size=cfg.patch_size, in_chans=cfg.in_chans, num_classes=cfg.num_classes,
                               embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, global_pool=cfg.global_pool,
                               precision=precision)


@pytest.mark.parametrize('gp', [False, True])
def test_oracle_matches_reference_features(gold, gp):
    tag = 'gp' if gp else 'cls'
    cfg = V.VitConfig(num_classes=3, global_pool=gp, **ENC)
    sd = V.init_vit_state_dict(cfg, seed=5)
    x = torch.from_numpy(gold['micro/x'])
    assert list(gold[f'micro/{tag}/keys']) == list(V.vit_state_dict_spec(cfg).keys())
    assert np.allclose(V.forward_features(sd, x, cfg).numpy(), gold[f'micro/{tag}/features'], atol=2e-6)
    assert np.allclose(V.forward(sd, x, cfg).numpy(), gold[f'micro/{tag}/logits'], atol=2e-6)
    mae_sd = R.init_state_dict(R.RefConfig(contrastive=True, **MICRO), seed=0)
    merged, missing, unexpected = V.handoff_from_mae(mae_sd, sd, cfg)
    assert sorted(missing) == list(gold[f'micro/{tag}/missing']) and sorted(unexpected) == list(gold[f'micro/{tag}/unexpected'])
    assert np.allclose(V.forward_features(merged, x, cfg).numpy(), gold[f'micro/{tag}/handoff_features'], atol=2e-6)


def test_oracle_vitb_features(gold):
    cfg = V.VitConfig(num_classes=2, global_pool=True, **VITB)
    sd = V.init_vit_state_dict(cfg, seed=7)
    xb, _ = R.synthetic_views((2, 4, 96, 96, 96), seed=1234)
    got = V.forward_features(sd, xb, cfg).numpy()
    assert np.abs(got - gold['vitb/features']).max() < 1e-4 * np.abs(gold['vitb/features']).max()


@pytest.mark.parametrize('gp', [False, True])
def test_module_keys_and_checkpoint_handoff(gold, gp):
    """load_state_dict(strict=False) of a pre-training checkpoint leaves exactly the keys the reference's script
    asserts on (extract_ssl_features.py:132-135)."""
    from vit_ae_plus_plus_amd.model.model_utils.vit_helpers import interpolate_pos_embed
    tag = 'gp' if gp else 'cls'
    cfg = V.VitConfig(num_classes=3, global_pool=gp, **ENC)
    m = _module(cfg)
    assert list(m.state_dict().keys()) == list(gold[f'micro/{tag}/keys'])
    assert float(m.head.weight.abs().sum()) == 0.0 and float(m.blocks[0].attn.qkv.weight.abs().sum()) > 0
    ck = dict(R.init_state_dict(R.RefConfig(contrastive=True, **MICRO), seed=0))
    interpolate_pos_embed(m, ck)
    msg = m.load_state_dict(ck, strict=False)
    assert sorted(msg.missing_keys) == list(gold[f'micro/{tag}/missing'])
    assert sorted(msg.unexpected_keys) == list(gold[f'micro/{tag}/unexpected'])
    expect = {'head.weight', 'head.bias'} | ({'fc_norm.weight', 'fc_norm.bias'} if gp else set())
    assert set(msg.missing_keys) == expect


def test_unsupported_variants_say_so():
    from vit_ae_plus_plus_amd.model.vit import VisionTransformer3D, VisionTransformer3DContrastive
    with pytest.raises(NotImplementedError):
        VisionTransformer3D(volume_size=16, patch_size=4, in_chans=1, embed_dim=24, depth=1, num_heads=2, distilled=True)
    with pytest.raises(NotImplementedError):
        VisionTransformer3DContrastive(volume_size=16, patch_size=4)
    from vit_ae_plus_plus_amd._abi import VitaeError
    m = VisionTransformer3D(volume_size=16, patch_size=4, in_chans=1, num_classes=0, embed_dim=24, depth=1, num_heads=2)
    with pytest.raises(VitaeError):          # no CPU fallback
        m.eval().forward_features(torch.zeros(1, 1, 16, 16, 16))


# ----------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize('gp', [False, True])
def test_hip_features_micro(gold, gp):
    tag = 'gp' if gp else 'cls'
    cfg = V.VitConfig(num_classes=3, global_pool=gp, **ENC)
    m = _module(cfg).cuda().eval()
    m.load_state_dict(V.init_vit_state_dict(cfg, seed=5))
    x = torch.from_numpy(gold['micro/x']).cuda()
    with torch.no_grad():
        f, y = m.forward_features(x), m(x)
    ref_f, ref_y = gold[f'micro/{tag}/features'], gold[f'micro/{tag}/logits']
    assert np.abs(f.cpu().numpy() - ref_f).max() < 1e-4 * np.abs(ref_f).max()      # fp32 tolerance of BASELINE north_star
    assert np.abs(y.cpu().numpy() - ref_y).max() < 1e-4 * max(1.0, np.abs(ref_y).max())
    # the checkpoint hand-off end to end
    from vit_ae_plus_plus_amd.model.model_utils.vit_helpers import interpolate_pos_embed
    ck = dict(R.init_state_dict(R.RefConfig(contrastive=True, **MICRO), seed=0))
    interpolate_pos_embed(m, ck)
    m.load_state_dict(ck, strict=False)
    with torch.no_grad():
        f2 = m.forward_features(x)
    ref2 = gold[f'micro/{tag}/handoff_features']
    assert np.abs(f2.cpu().numpy() - ref2).max() < 1e-4 * np.abs(ref2).max()


@pytest.mark.gpu
@pytest.mark.parametrize('precision,tol', [('fp32', 1e-4), ('fp32x3', 1e-4), ('bf16', 3e-2)])   # fp32x3: split-operand GEMMs, the fp32 bound
def test_hip_features_vitb(gold, precision, tol):
    """ViT-B/16 on 96^3 x 4ch (N = 217 tokens): fp32 mode inside the 1e-4 of the north star, bf16 (the reference runs
    this call under autocast) to bf16 round-off."""
    cfg = V.VitConfig(num_classes=2, global_pool=True, **VITB)
    m = _module(cfg, precision).cuda().eval()
    m.load_state_dict(V.init_vit_state_dict(cfg, seed=7))
    xb, _ = R.synthetic_views((2, 4, 96, 96, 96), seed=1234)
    with torch.no_grad():
        f = m.forward_features(xb.cuda())
    ref = gold['vitb/features']
    err = np.abs(f.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < tol, err
    if precision == 'bf16':
        assert m._encoder.act16            # the LDS-DMA bf16 path, not the generic launcher


@pytest.mark.gpu
def test_generate_features_writes_reference_files(tmp_path, gold):
    from vit_ae_plus_plus_amd.utils.feature_extraction import generate_features
    cfg = V.VitConfig(num_classes=3, global_pool=True, **ENC)
    m = _module(cfg).cuda()
    m.load_state_dict(V.init_vit_state_dict(cfg, seed=5))
    x = torch.from_numpy(gold['micro/x'])
    loader = [(x[:2], torch.tensor([0., 1.])), (x[2:], torch.tensor([1.]))]
    generate_features(loader, m, torch.device('cuda'), str(tmp_path))
    f, l = np.load(tmp_path / 'features.npy'), np.load(tmp_path / 'gt_labels.npy')
    assert f.shape == (3, cfg.embed_dim) and list(l) == [0., 1., 1.]
    assert np.abs(f - gold['micro/gp/features']).max() < 1e-4 * np.abs(gold['micro/gp/features']).max()
