"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's encoder-only
ViT used for feature extraction after pre-training.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.

Follows
  * ``VisionTransformer3D.__init__``          model/vit.py:157-226  (module tree / state-dict keys)
  * ``VisionTransformer3D.forward_features``  model/vit.py:265-284
  * ``VisionTransformer3D.forward``           model/vit.py:286-297  (non-distilled: head(features))
  * the checkpoint hand-off of post_training_utils/extract_ssl_features.py:111-135
    (drop mismatching head, interpolate_pos_embed, load_state_dict(strict=False), expected missing keys)
and is pinned by tests/golden/vit_features.npz, produced by running the reference's own
VisionTransformer3D on the same weights and inputs (oracle/gen_golden.py: gen_vit_features).

Out of scope, as in the product: distilled (DeiT) token/head, ``representation_size`` pre-logits,
dropout / stochastic depth > 0 (all identities at the reference's feature-extraction settings).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

import torch

from . import mae_ref as R


@dataclass
class VitConfig:
    """Constructor arguments of model/vit.py:157-160 that change the arithmetic."""
    volume_size: Tuple[int, int, int] = (96, 96, 96)
    patch_size: int = 16
    in_chans: int = 4
    num_classes: int = 2
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    global_pool: bool = False
    ln_eps: float = 1e-6          # model_factory.py:21 passes partial(nn.LayerNorm, eps=1e-6)

    def __post_init__(self):
        self.volume_size = R._triple(self.volume_size)

    @property
    def grid(self):
        return tuple(v // self.patch_size for v in self.volume_size)

    @property
    def num_patches(self):
        g = self.grid
        return g[0] * g[1] * g[2]


def vit_state_dict_spec(cfg: VitConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Keys and shapes in module-registration order (model/vit.py:188-224)."""
    D, p, C = cfg.embed_dim, cfg.patch_size, cfg.in_chans
    H = int(D * cfg.mlp_ratio)
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    spec['cls_token'] = (1, 1, D)
    spec['pos_embed'] = (1, cfg.num_patches + 1, D)
    spec['patch_embed.proj.weight'] = (D, C, p, p, p)
    spec['patch_embed.proj.bias'] = (D,)
    for i in range(cfg.depth):
        b = f'blocks.{i}.'
        spec[b + 'norm1.weight'] = (D,); spec[b + 'norm1.bias'] = (D,)
        spec[b + 'attn.qkv.weight'] = (3 * D, D); spec[b + 'attn.qkv.bias'] = (3 * D,)
        spec[b + 'attn.proj.weight'] = (D, D); spec[b + 'attn.proj.bias'] = (D,)
        spec[b + 'norm2.weight'] = (D,); spec[b + 'norm2.bias'] = (D,)
        spec[b + 'mlp.fc1.weight'] = (H, D); spec[b + 'mlp.fc1.bias'] = (H,)
        spec[b + 'mlp.fc2.weight'] = (D, H); spec[b + 'mlp.fc2.bias'] = (D,)
    if not cfg.global_pool:                      # vit.py:218-221: fc_norm replaces norm
        spec['norm.weight'] = (D,); spec['norm.bias'] = (D,)
    if cfg.num_classes > 0:                      # vit.py:214
        spec['head.weight'] = (cfg.num_classes, D); spec['head.bias'] = (cfg.num_classes,)
    if cfg.global_pool:
        spec['fc_norm.weight'] = (D,); spec['fc_norm.bias'] = (D,)
    return spec


def init_vit_state_dict(cfg: VitConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic stand-in weights (a fixture generator, NOT the reference's init distribution):
    every tensor N(0, 0.05) except LayerNorm weights 1 + N(0, 0.05), so no term of the forward is
    trivially zero (the reference zero-initialises ``head``)."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in vit_state_dict_spec(cfg).items():
        t = torch.randn(shp, generator=g) * 0.05
        if 'norm' in k and k.endswith('.weight'):
            t = t + 1.0
        sd[k] = t
    return sd


def forward_features(sd: Dict[str, torch.Tensor], x: torch.Tensor, cfg: VitConfig) -> torch.Tensor:
    """model/vit.py:265-284."""
    B = x.shape[0]
    t = R.patch_embed(x, sd, cfg.patch_size)                              # :267
    t = torch.cat([sd['cls_token'].expand(B, -1, -1), t], dim=1)          # :269-270
    t = t + sd['pos_embed']                                               # :271 (pos_drop p=0)
    for i in range(cfg.depth):                                            # :274-275
        t = R.block(t, sd, f'blocks.{i}.', cfg.num_heads, cfg.ln_eps)
    if cfg.global_pool:                                                   # :277-279
        return R._ln(t[:, 1:, :].mean(dim=1), sd, 'fc_norm.', cfg.ln_eps)
    return R._ln(t, sd, 'norm.', cfg.ln_eps)[:, 0]                        # :281-282


def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, cfg: VitConfig) -> torch.Tensor:
    """model/vit.py:286-297 without the distillation head."""
    f = forward_features(sd, x, cfg)
    if cfg.num_classes > 0:
        f = torch.nn.functional.linear(f, sd['head.weight'], sd['head.bias'])
    return f


def handoff_from_mae(mae_sd: Dict[str, torch.Tensor], vit_sd: Dict[str, torch.Tensor], cfg: VitConfig):
    """post_training_utils/extract_ssl_features.py:113-135: what ``load_state_dict(strict=False)`` of an MAE
    checkpoint into the encoder-only model does.  Returns (merged state dict, missing keys, unexpected keys)."""
    ck = dict(mae_sd)
    for k in ('head.weight', 'head.bias'):
        if k in ck and k in vit_sd and ck[k].shape != vit_sd[k].shape:
            del ck[k]
    merged = OrderedDict((k, v.clone()) for k, v in vit_sd.items())
    for k, v in ck.items():
        if k in merged:
            assert merged[k].shape == v.shape, (k, merged[k].shape, v.shape)
            merged[k] = v.clone()
    missing = [k for k in vit_sd if k not in ck]
    unexpected = [k for k in ck if k not in vit_sd]
    return merged, missing, unexpected
