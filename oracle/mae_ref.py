"""Plain-PyTorch CPU restatement of the reference's 3D ViT-MAE (+contrastive) hot path.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Functional style: the model is an ordered
``state_dict`` (same keys/shapes as the reference modules) plus a ``RefConfig``.
Every function cites the reference ``file:line`` (relative to ``/root/reference``) it restates.
Works in fp32 (the reference's precision) or fp64 (tighter yardstick for GPU tests).

Parity pinned: ``tests/test_oracle_golden.py`` compares this file with fixtures produced by the
imported reference (``oracle/gen_golden.py``).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- config
def _triple(v) -> Tuple[int, int, int]:
    """model/vit.py:48-49 ``traid``."""
    if isinstance(v, (tuple, list)):
        assert len(v) == 3
        return tuple(int(i) for i in v)
    return (int(v),) * 3


@dataclass
class RefConfig:
    """Constructor arguments of ``MaskedAutoencoderViT`` / ``ContrastiveMAEViT``
    (model/vit_autoenc.py:18-21, 242-245)."""
    volume_size: Tuple[int, int, int] = (96, 96, 96)
    patch_size: int = 16
    in_chans: int = 4
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    decoder_embed_dim: int = 512
    decoder_depth: int = 8
    decoder_num_heads: int = 16
    mlp_ratio: float = 4.0
    contrastive: bool = False
    norm_pix_loss: bool = False
    ln_eps: float = 1e-6  # partial(nn.LayerNorm, eps=1e-6), vit_autoenc.py:292,300,308

    def __post_init__(self):
        self.volume_size = _triple(self.volume_size)
        p = int(self.patch_size)
        assert all(v % p == 0 for v in self.volume_size)

    @property
    def grid(self) -> Tuple[int, int, int]:  # model/vit.py:61
        p = self.patch_size
        return tuple(v // p for v in self.volume_size)

    @property
    def num_patches(self) -> int:  # model/vit.py:62
        g = self.grid
        return g[0] * g[1] * g[2]

    @property
    def patch_dim(self) -> int:  # vit_autoenc.py:53
        return self.patch_size ** 3 * self.in_chans

    def len_keep(self, mask_ratio: float) -> int:  # vit_autoenc.py:137
        return int(self.num_patches * (1 - mask_ratio))


def vit_base_cfg(**kw) -> RefConfig:  # vit_autoenc.py:296-301 / 304-309
    return RefConfig(embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512,
                     decoder_depth=8, decoder_num_heads=16, **kw)


def vit_large_cfg(**kw) -> RefConfig:  # vit_autoenc.py:288-293
    return RefConfig(embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                     decoder_depth=8, decoder_num_heads=16, **kw)


# ----------------------------------------------------------------------------- fixed tables
def sincos_1d(dim: int, pos: np.ndarray) -> np.ndarray:
    """model/model_utils/vit_helpers.py:48-70 (float64)."""
    assert dim % 2 == 0
    omega = np.arange(dim // 2, dtype=np.float64)
    omega /= dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum('m,d->md', pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed_3d(embed_dim: int, grid, cls_token: bool = True) -> np.ndarray:
    """model/model_utils/vit_helpers.py:13-45, generalised to a non-cubic grid.

    The reference feeds ``np.meshgrid(grid_l, grid_h, grid_w)`` (default 'xy' indexing,
    vit_helpers.py:22) so for token (l, h, w) (C-order flattening, model/vit.py:74) the three
    column blocks encode [h | l | w].  Column split: res = D//3 rounded up to even,
    last block D - 2*res (vit_helpers.py:36-42).
    """
    gl, gh, gw = _triple(grid)
    assert embed_dim % 2 == 0
    res = embed_dim // 3
    if res % 2 != 0:
        res += 1
    last = embed_dim - 2 * res
    ll, hh, ww = np.meshgrid(np.arange(gl), np.arange(gh), np.arange(gw), indexing='ij')
    emb = np.concatenate([sincos_1d(res, hh.astype(np.float32)),
                          sincos_1d(res, ll.astype(np.float32)),
                          sincos_1d(last, ww.astype(np.float32))], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def gaussian_taps(sigma: float = 2) -> torch.Tensor:
    """model/model_utils/gaussian_filter.py:5-13: ks=int(5*sigma) made odd; taps sampled at
    ``linspace(-ks//2, ks//2+1, ks)`` (for sigma=2: linspace(-6, 6, 11), step 1.2)."""
    ks = int(sigma * 5)
    if ks % 2 == 0:
        ks += 1
    ts = torch.linspace(-ks // 2, ks // 2 + 1, ks)
    g = torch.exp(-(ts / sigma) ** 2 / 2)
    return g / g.sum()


def gaussian_blur3d(vol: torch.Tensor, sigma: float = 2) -> torch.Tensor:
    """gaussian_filter.py:16-26: per channel dense conv3d with outer(k,k,k)/sum, zero pad."""
    k = gaussian_taps(sigma).to(vol.dtype)
    k3 = torch.einsum('i,j,k->ijk', k, k, k)
    k3 = k3 / k3.sum()
    B, C = vol.shape[:2]
    out = F.conv3d(vol.reshape(B * C, 1, *vol.shape[2:]), k3[None, None], padding=len(k) // 2)
    return out.reshape(vol.shape)


def sobel_kernels(dtype=torch.float32) -> torch.Tensor:
    """model/model_utils/sobel_filter.py:12-31 -> weight [3,1,3,3,3]."""
    s = torch.tensor([1., 2., 1.], dtype=dtype)
    d = torch.tensor([1., 0., -1.], dtype=dtype)
    k0 = torch.einsum('i,j,k->ijk', s, s, d)       # derivative along last axis, [1,0,-1]
    k1 = torch.einsum('i,j,k->ijk', s, -d, s)      # along middle axis, [-1,0,1]
    k2 = torch.einsum('i,j,k->ijk', -d, s, s)      # along first axis, [-1,0,1]
    return torch.stack([k0, k1, k2])[:, None]


def sobel_magnitude(vol: torch.Tensor) -> torch.Tensor:
    """sobel_filter.py:37-45: per channel sqrt(gx^2+gy^2+gz^2), summed over channels
    -> [B, Lz, Hy, Wx].  Zero padding 1, bias 0."""
    B, C = vol.shape[:2]
    g = F.conv3d(vol.reshape(B * C, 1, *vol.shape[2:]), sobel_kernels(vol.dtype), padding=1)
    mag = torch.sqrt((g ** 2).sum(dim=1))
    return mag.reshape(B, C, *vol.shape[2:]).sum(dim=1)


# ----------------------------------------------------------------------------- permutations
def patchify(vol: torch.Tensor, p: int) -> torch.Tensor:
    """vit_autoenc.py:100-113 (einsum 'nclrhpwq->nlhwrpqc'), generalised to non-cubic."""
    B, C, Lz, Hy, Wx = vol.shape
    l, h, w = Lz // p, Hy // p, Wx // p
    x = vol.reshape(B, C, l, p, h, p, w, p)
    x = x.permute(0, 2, 4, 6, 3, 5, 7, 1)
    return x.reshape(B, l * h * w, p * p * p * C)


def unpatchify(x: torch.Tensor, p: int, grid) -> torch.Tensor:
    """vit_autoenc.py:115-128 (einsum 'nlhwrpqc->nclrhpwq')."""
    l, h, w = grid
    B = x.shape[0]
    x = x.reshape(B, l, h, w, p, p, p, -1)
    x = x.permute(0, 7, 1, 4, 2, 5, 3, 6)
    return x.reshape(B, -1, l * p, h * p, w * p)


def masking_from_noise(noise: torch.Tensor, len_keep: int):
    """vit_autoenc.py:141-153 given the ``noise`` of :139."""
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :len_keep]
    mask = torch.ones_like(noise)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return ids_keep, ids_restore, mask


# ----------------------------------------------------------------------------- transformer
def _attention(x, sd, pre, heads):
    """model/vit.py:112-124."""
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[pre + 'qkv.weight'], sd[pre + 'qkv.bias'])
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * hd ** -0.5
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[pre + 'proj.weight'], sd[pre + 'proj.bias'])


def _mlp(x, sd, pre):
    """model/vit.py:90-96 (exact-erf GELU, dropout p=0)."""
    h = F.gelu(F.linear(x, sd[pre + 'fc1.weight'], sd[pre + 'fc1.bias']))
    return F.linear(h, sd[pre + 'fc2.weight'], sd[pre + 'fc2.bias'])


def _ln(x, sd, pre, eps):
    return F.layer_norm(x, x.shape[-1:], sd[pre + 'weight'], sd[pre + 'bias'], eps)


def block(x, sd, pre, heads, eps):
    """model/vit.py:139-144 (pre-LN residual block)."""
    x = x + _attention(_ln(x, sd, pre + 'norm1.', eps), sd, pre + 'attn.', heads)
    x = x + _mlp(_ln(x, sd, pre + 'norm2.', eps), sd, pre + 'mlp.')
    return x


def patch_embed(x, sd, p):
    """model/vit.py:68-76: Conv3d(k=p, s=p) -> flatten(2).transpose(1,2)."""
    y = F.conv3d(x, sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias'], stride=p)
    return y.flatten(2).transpose(1, 2)


def forward_encoder(sd, x, noise, cfg: RefConfig, mask_ratio: float, trace: Optional[dict] = None):
    """vit_autoenc.py:157-177 with the masking noise of :139 passed in."""
    t = trace if trace is not None else {}
    pe = patch_embed(x, sd, cfg.patch_size)
    t['patch_embed'] = pe
    h = pe + sd['pos_embed'][:, 1:, :]
    keep = cfg.len_keep(mask_ratio)
    ids_keep, ids_restore, mask = masking_from_noise(noise, keep)
    D = h.shape[-1]
    xm = torch.gather(h, 1, ids_keep.unsqueeze(-1).repeat(1, 1, D))
    t['x_masked'] = xm
    cls = (sd['cls_token'] + sd['pos_embed'][:, :1, :]).expand(xm.shape[0], -1, -1)
    h = torch.cat((cls, xm), dim=1)
    for i in range(cfg.depth):
        h = block(h, sd, f'blocks.{i}.', cfg.num_heads, cfg.ln_eps)
        t[f'enc_block{i}'] = h
    h = _ln(h, sd, 'norm.', cfg.ln_eps)
    return h, mask, ids_restore


def forward_decoder(sd, latent, ids_restore, cfg: RefConfig, trace: Optional[dict] = None):
    """vit_autoenc.py:179-203."""
    t = trace if trace is not None else {}
    x = F.linear(latent, sd['decoder_embed.weight'], sd['decoder_embed.bias'])
    B, Ne, Dd = x.shape
    L = ids_restore.shape[1]
    mask_tokens = sd['mask_token'].repeat(B, L + 1 - Ne, 1)
    x_ = torch.cat([x[:, 1:, :], mask_tokens], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).repeat(1, 1, Dd))
    x = torch.cat([x[:, :1, :], x_], dim=1)
    x = x + sd['decoder_pos_embed']
    t['decoder_in'] = x
    for i in range(cfg.decoder_depth):
        x = block(x, sd, f'decoder_blocks.{i}.', cfg.decoder_num_heads, cfg.ln_eps)
        t[f'dec_block{i}'] = x
    x = _ln(x, sd, 'decoder_norm.', cfg.ln_eps)
    x = F.linear(x, sd['decoder_pred.weight'], sd['decoder_pred.bias'])
    return x[:, 1:, :]


def loss_terms(imgs, pred, mask, cfg: RefConfig, edge_map_weight: float = 0.0,
               perceptual_weight: float = 0.0, trace: Optional[dict] = None):
    """vit_autoenc.py:205-232 -> [loss, raw_edge_mse, recon, percep].

    The perceptual term is a zero-gradient logging value with weight 0 by default;
    it is restated as the constant 0.
    """
    t = trace if trace is not None else {}
    p, grid = cfg.patch_size, cfg.grid
    target = patchify(imgs, p)
    if cfg.norm_pix_loss:  # :212-215 (unbiased var)
        mean = target.mean(dim=-1, keepdim=True)
        var = target.var(dim=-1, keepdim=True)
        target = (target - mean) / (var + 1.e-6) ** .5
    t['target'] = target
    pred_vol, target_vol = unpatchify(pred, p, grid), unpatchify(target, p, grid)
    blurred = gaussian_blur3d(target_vol, 2)
    e_pred, e_tgt = sobel_magnitude(pred_vol), sobel_magnitude(blurred)
    t['blurred'], t['edge_pred'], t['edge_target'] = blurred, e_pred, e_tgt
    raw_edge = F.mse_loss(e_pred, e_tgt, reduction='mean')
    edge = edge_map_weight * F.mse_loss(e_pred, e_tgt, reduction='mean')
    recon = ((pred - target) ** 2).mean(dim=-1)
    recon = (recon * mask).sum() / mask.sum()
    percep = torch.zeros((), dtype=pred.dtype) * perceptual_weight
    loss = edge + recon + percep
    return [loss, raw_edge, recon, percep]


def predictor(sd, z, bn_state: Optional[dict] = None, training: bool = True, momentum=0.1, eps=1e-5):
    """vit_autoenc.py:263-268: Linear(no bias) -> BatchNorm1d -> ReLU -> Linear(+bias).
    ``bn_state`` holds running_mean / running_var / num_batches_tracked and is updated in place
    in training mode like nn.BatchNorm1d."""
    h = F.linear(z, sd['predictor.0.weight'])
    if bn_state is None:
        bn_state = {'running_mean': sd['predictor.1.running_mean'].clone(),
                    'running_var': sd['predictor.1.running_var'].clone(),
                    'num_batches_tracked': sd['predictor.1.num_batches_tracked'].clone()}
    if training:
        bn_state['num_batches_tracked'] += 1
    h = F.batch_norm(h, bn_state['running_mean'], bn_state['running_var'],
                     sd['predictor.1.weight'], sd['predictor.1.bias'], training, momentum, eps)
    h = F.relu(h)
    return F.linear(h, sd['predictor.3.weight'], sd['predictor.3.bias'])


def mae_forward(sd, x, noise, cfg: RefConfig, mask_ratio=0.75, edge_map_weight=0.0, trace=None):
    """vit_autoenc.py:234-238 -> (loss_list, pred, mask)."""
    latent, mask, ids_restore = forward_encoder(sd, x, noise, cfg, mask_ratio, trace)
    if trace is not None:
        trace['latent'], trace['mask'], trace['ids_restore'] = latent, mask, ids_restore
    pred = forward_decoder(sd, latent, ids_restore, cfg, trace)
    loss = loss_terms(x, pred, mask, cfg, edge_map_weight, trace=trace)
    return loss, pred, mask


def contr_forward(sd, view1, view2, noise1, noise2, cfg: RefConfig, mask_ratio=0.75,
                  edge_map_weight=0.0, bn_state=None, training=True, trace=None):
    """vit_autoenc.py:270-285 -> (loss_list, pred, mask, p1, p2, z1.detach, z2.detach)."""
    latent1, mask, ids_restore = forward_encoder(sd, view1, noise1, cfg, mask_ratio, trace)
    if trace is not None:
        trace['latent'], trace['mask'], trace['ids_restore'] = latent1, mask, ids_restore
    pred = forward_decoder(sd, latent1, ids_restore, cfg, trace)
    loss = loss_terms(view1, pred, mask, cfg, edge_map_weight, trace=trace)
    latent2, _, _ = forward_encoder(sd, view2, noise2, cfg, mask_ratio)
    z1 = latent1.reshape(-1, latent1.shape[2])
    z2 = latent2.reshape(-1, latent2.shape[2])
    if bn_state is None:
        bn_state = {k: sd['predictor.1.' + k].clone()
                    for k in ('running_mean', 'running_var', 'num_batches_tracked')}
    p1 = predictor(sd, z1, bn_state, training)
    p2 = predictor(sd, z2, bn_state, training)
    if trace is not None:
        trace['latent2'] = latent2
        trace['bn_state'] = bn_state
    return loss, pred, mask, p1, p2, z1.detach(), z2.detach()


def contrastive_loss(p1, p2, z1, z2, contr_weight: float):
    """utils/train_one_epoch.py:32,113-114 (nn.CosineSimilarity(dim=1), eps 1e-8)."""
    c = lambda a, b: F.cosine_similarity(a, b, dim=1, eps=1e-8)
    return contr_weight * (-(c(p1, z2).mean() + c(p2, z1).mean()) * 0.5)


# ----------------------------------------------------------------------------- parameters
FROZEN_KEYS = ('pos_embed', 'decoder_pos_embed', 'sobel_filter3D.sobel_filter.weight',
               'sobel_filter3D.sobel_filter.bias')
BUFFER_SUFFIXES = ('running_mean', 'running_var', 'num_batches_tracked')


def state_dict_spec(cfg: RefConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Key -> shape, in the reference's registration order."""
    D, Dd, p, C = cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_size, cfg.in_chans
    L, P = cfg.num_patches, cfg.patch_dim
    H, Hd = int(D * cfg.mlp_ratio), int(Dd * cfg.mlp_ratio)
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s['cls_token'] = (1, 1, D)
    s['pos_embed'] = (1, L + 1, D)
    s['mask_token'] = (1, 1, Dd)
    s['decoder_pos_embed'] = (1, L + 1, Dd)
    s['patch_embed.proj.weight'] = (D, C, p, p, p)
    s['patch_embed.proj.bias'] = (D,)

    def blk(pre, d, h):
        s[pre + 'norm1.weight'] = (d,); s[pre + 'norm1.bias'] = (d,)
        s[pre + 'attn.qkv.weight'] = (3 * d, d); s[pre + 'attn.qkv.bias'] = (3 * d,)
        s[pre + 'attn.proj.weight'] = (d, d); s[pre + 'attn.proj.bias'] = (d,)
        s[pre + 'norm2.weight'] = (d,); s[pre + 'norm2.bias'] = (d,)
        s[pre + 'mlp.fc1.weight'] = (h, d); s[pre + 'mlp.fc1.bias'] = (h,)
        s[pre + 'mlp.fc2.weight'] = (d, h); s[pre + 'mlp.fc2.bias'] = (d,)

    for i in range(cfg.depth):
        blk(f'blocks.{i}.', D, H)
    s['norm.weight'] = (D,); s['norm.bias'] = (D,)
    s['decoder_embed.weight'] = (Dd, D); s['decoder_embed.bias'] = (Dd,)
    for i in range(cfg.decoder_depth):
        blk(f'decoder_blocks.{i}.', Dd, Hd)
    s['decoder_norm.weight'] = (Dd,); s['decoder_norm.bias'] = (Dd,)
    s['decoder_pred.weight'] = (P, Dd); s['decoder_pred.bias'] = (P,)
    s['sobel_filter3D.sobel_filter.weight'] = (3, 1, 3, 3, 3)
    s['sobel_filter3D.sobel_filter.bias'] = (3,)
    if cfg.contrastive:
        s['predictor.0.weight'] = (D, D)
        s['predictor.1.weight'] = (D,); s['predictor.1.bias'] = (D,)
        s['predictor.1.running_mean'] = (D,); s['predictor.1.running_var'] = (D,)
        s['predictor.1.num_batches_tracked'] = ()
        s['predictor.3.weight'] = (D, D); s['predictor.3.bias'] = (D,)
    return s


def is_trainable(key: str) -> bool:
    return key not in FROZEN_KEYS and not key.endswith(BUFFER_SUFFIXES)


def init_state_dict(cfg: RefConfig, seed: int = 0, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic CPU init with the reference's distributions (vit_autoenc.py:65-98):
    sin-cos tables; xavier-uniform for every Linear weight and for the patch-embed weight viewed
    as [D, C*p^3]; zeros for Linear biases; LayerNorm (1, 0); cls/mask tokens N(0, .02);
    patch-embed bias keeps nn.Conv3d's default U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BatchNorm
    (1, 0) with running (0, 1).  The random *stream* is this function's own (weights always
    travel by state_dict)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def xavier(shape2d):
        fan_out, fan_in = shape2d
        a = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape2d, generator=g, dtype=torch.float64) * 2 - 1) * a

    for k, shp in state_dict_spec(cfg).items():
        if k == 'pos_embed':
            v = torch.from_numpy(sincos_pos_embed_3d(cfg.embed_dim, cfg.grid, True))[None]
        elif k == 'decoder_pos_embed':
            v = torch.from_numpy(sincos_pos_embed_3d(cfg.decoder_embed_dim, cfg.grid, True))[None]
        elif k in ('cls_token', 'mask_token'):
            v = torch.randn(shp, generator=g, dtype=torch.float64) * 0.02
        elif k == 'patch_embed.proj.weight':
            v = xavier((shp[0], int(np.prod(shp[1:])))).reshape(shp)
        elif k == 'patch_embed.proj.bias':
            bound = 1.0 / math.sqrt(cfg.patch_dim)
            v = (torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * bound
        elif k == 'sobel_filter3D.sobel_filter.weight':
            v = sobel_kernels(torch.float64)
        elif k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.long)
            continue
        elif k.endswith('running_var'):
            v = torch.ones(shp, dtype=torch.float64)
        elif k.endswith('.weight') and len(shp) == 2:
            v = xavier(shp)
        elif k.endswith('.weight'):   # LayerNorm / BatchNorm scale
            v = torch.ones(shp, dtype=torch.float64)
        else:                           # biases, running_mean, sobel bias
            v = torch.zeros(shp, dtype=torch.float64)
        sd[k] = v.to(dtype).contiguous()
    return sd


def param_groups(sd: Dict[str, torch.Tensor], weight_decay: float):
    """timm==0.5.4 ``optim_factory.add_weight_decay`` (call site
    k_fold_training_scripts/k_fold_cross_valid_combined_brats.py:168): frozen params skipped;
    no decay iff ndim == 1 or name ends with '.bias'.  (Third-party, parity unpinned — SURVEY §8c.)"""
    decay, no_decay = [], []
    for k, v in sd.items():
        if not is_trainable(k):
            continue
        if v.ndim <= 1 or k.endswith('.bias'):
            no_decay.append(k)
        else:
            decay.append(k)
    return [{'names': no_decay, 'weight_decay': 0.0}, {'names': decay, 'weight_decay': weight_decay}]


def make_leaf_params(sd: Dict[str, torch.Tensor], dtype=None) -> "OrderedDict[str, torch.Tensor]":
    """Clone a state dict into autograd leaves (trainable keys get requires_grad)."""
    out = OrderedDict()
    for k, v in sd.items():
        t = v.detach().clone()
        if dtype is not None and t.is_floating_point():
            t = t.to(dtype)
        if is_trainable(k):
            t.requires_grad_(True)
        out[k] = t
    return out


# ----------------------------------------------------------------------------- synthetic data
def synthetic_views(shape: Sequence[int], seed: int):
    """SURVEY §8d synthetic inputs: view2 = N(0,1); view1 = per-channel z-score of
    view2 + 0.1*N(0,1) (mimics tio.RandomNoise(std=.1) + _normalize_data,
    k_fold_cross_valid_combined_brats.py:93-97, dataset/egd.py:44-47)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    view2 = torch.randn(*shape, generator=g)
    v1 = view2 + 0.1 * torch.randn(*shape, generator=g)
    dims = (2, 3, 4)
    view1 = (v1 - v1.mean(dim=dims, keepdim=True)) / v1.std(dim=dims, keepdim=True)
    return view1.contiguous(), view2.contiguous()


def masking_noise(batch: int, num_patches: int, seed: int, n: int = 2):
    """``n`` independent U[0,1) [B, L] noises from a CPU generator (vit_autoenc.py:139)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    return [torch.rand(batch, num_patches, generator=g) for _ in range(n)]
