"""Building blocks of the 3-D ViT with the reference's names and parameter layout
(reference: model/vit.py:52-144).

In this package the modules are *parameter containers*: they own tensors under the reference's
state-dict keys (``proj.weight``, ``attn.qkv.weight``, ``mlp.fc1.bias`` ...) so checkpoints
round-trip, while the arithmetic of a whole model is sequenced by ``HipMAEEngine`` over
libvitae_hip.so.  Their stand-alone ``forward`` methods run the same HIP kernels op by op
(inference only: used by feature extraction, not by the training hot path).
"""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn

from .._abi import CONSTS, VitaeError, lib


def traid(t):
    return t if isinstance(t, tuple) else (t, t, t)


def _stream(x):
    return torch.cuda.current_stream(x.device).cuda_stream


def _require_hip(x, what):
    if not x.is_cuda:
        raise VitaeError(f'{what}: input is on {x.device}; this package computes on MI355X only '
                         f'(no CPU fallback).')
    if torch.is_grad_enabled() and x.requires_grad:
        raise VitaeError(f'{what}: stand-alone module forward is inference-only; training goes through '
                         f'MaskedAutoencoderViT / ContrastiveMAEViT')


def hip_linear(x, weight, bias, epi=0, residual=None, precision=0):
    """y = x W^T + b via vitae_linear_fwd (inference helper)."""
    K = x.shape[-1]
    N = weight.shape[0]
    x2 = x.reshape(-1, K).contiguous().float()
    M = x2.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    aux = torch.empty_like(y) if epi == CONSTS['VITAE_EPI_GELU'] else None
    split = lib.vitae_gemm_pick_split_k(M, N, K) if epi == 0 else 1
    ws = torch.empty(split * M * N, dtype=torch.float32, device=x.device) if split > 1 else None
    lib.vitae_linear_fwd(precision, x2.data_ptr(), weight.data_ptr(), None if bias is None else bias.data_ptr(),
                         y.data_ptr(), M, N, K, epi, None if aux is None else aux.data_ptr(),
                         None if residual is None else residual.data_ptr(), split,
                         None if ws is None else ws.data_ptr(), _stream(x))
    return y.reshape(*x.shape[:-1], N)


def hip_layernorm(x, weight, bias, eps):
    D = x.shape[-1]
    x2 = x.reshape(-1, D).contiguous().float()
    M = x2.shape[0]
    y = torch.empty_like(x2)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    lib.vitae_layernorm_fwd(x2.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), None, mean.data_ptr(),
                            rstd.data_ptr(), M, D, eps, _stream(x))
    return y.reshape(x.shape)


class PatchEmbed3D(nn.Module):
    """3-D volume to patch embedding (reference model/vit.py:52-76): Conv3d(k = s = patch)."""

    def __init__(self, volume_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        volume_size, patch_size = traid(volume_size), traid(patch_size)
        self.volume_size, self.patch_size = volume_size, patch_size
        self.grid_size = tuple(v // p for v, p in zip(volume_size, patch_size))
        self.num_patches = self.grid_size[0] * self.grid_size[1] * self.grid_size[2]
        self.flatten = flatten
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        B, C, L, H, W = x.shape
        assert (L, H, W) == tuple(self.volume_size), \
            f"Volume image size ({L}*{H}*{W}) doesn't match model ({self.volume_size[0]}*{self.volume_size[1]}*{self.volume_size[2]})."
        _require_hip(x, 'PatchEmbed3D')
        p = self.patch_size[0]
        if self.patch_size != (p, p, p):
            raise VitaeError('anisotropic patches are not supported')
        n = self.num_patches
        ids = torch.arange(n, dtype=torch.int32, device=x.device).repeat(B, 1).contiguous()
        xc = x.contiguous().float()
        rows = torch.empty(B * n, C * p ** 3, dtype=torch.float32, device=x.device)
        lib.vitae_gather_patches(xc.data_ptr(), ids.data_ptr(), rows.data_ptr(), None, B, C, L, H, W, p, n, _stream(x))
        w = self.proj.weight.reshape(self.proj.weight.shape[0], -1)
        y = hip_linear(rows, w, self.proj.bias).reshape(B, n, -1)
        if not self.flatten:
            y = y.transpose(1, 2).reshape(B, -1, *self.grid_size)
        return self.norm(y) if not isinstance(self.norm, nn.Identity) else y


class Mlp3D(nn.Module):
    """fc1 -> exact GELU -> fc2 (reference model/vit.py:78-96; dropout p=0 is the identity)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if drop:
            raise VitaeError('dropout > 0 is not supported (the reference MAE path uses 0)')
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, residual=None):
        _require_hip(x, 'Mlp3D')
        h = hip_linear(x, self.fc1.weight, self.fc1.bias, epi=CONSTS['VITAE_EPI_GELU'])
        return hip_linear(h, self.fc2.weight, self.fc2.bias, residual=residual)


class Attention(nn.Module):
    """Multi-head self-attention (reference model/vit.py:100-124)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        if attn_drop or proj_drop:
            raise VitaeError('dropout > 0 is not supported (the reference MAE path uses 0)')
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, residual=None):
        _require_hip(x, 'Attention')
        B, N, C = x.shape
        qkv = hip_linear(x, self.qkv.weight, self.qkv.bias).contiguous()
        o = torch.empty(B, N, C, dtype=torch.float32, device=x.device)
        lse = torch.empty(B * self.num_heads * N, dtype=torch.float32, device=x.device)
        lib.vitae_sdpa_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, self.num_heads, C // self.num_heads,
                           _stream(x))
        return hip_linear(o, self.proj.weight, self.proj.bias, residual=residual)


class Block(nn.Module):
    """Pre-LN transformer block (reference model/vit.py:126-144)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp3D(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x):
        _require_hip(x, 'Block')
        x = x.contiguous().float()
        x = self.attn(hip_layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps), residual=x)
        return self.mlp(hip_layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps), residual=x)


def _init_vit_weights(module: nn.Module, name: str = '', head_bias: float = 0.):
    """The reference's default ('') init scheme (model/vit.py:14-46): Linear weights trunc-normal(std .02) with zero
    bias, the classifier head zero, LayerNorm (1, 0); convolutions keep PyTorch's default."""
    if isinstance(module, nn.Linear):
        if name.startswith('head'):
            nn.init.zeros_(module.weight)
            nn.init.constant_(module.bias, head_bias)
        else:
            nn.init.trunc_normal_(module.weight, std=.02)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
    elif isinstance(module, (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm2d)):
        nn.init.zeros_(module.bias)
        nn.init.ones_(module.weight)


class VisionTransformer3D(nn.Module):
    """Encoder-only 3-D ViT that consumes the pre-trained weights (reference model/vit.py:147-298): same constructor,
    state-dict keys and ``forward_features`` / ``forward`` results; the arithmetic runs on the HIP kernels through
    ``HipEncoder`` (inference: feature extraction, utils/feature_extraction.py).  Not covered: the DeiT distillation
    token/head, ``representation_size`` pre-logits, non-zero dropout / stochastic depth, and back-propagation
    (fine-tuning) — the constructor or the call says so instead of computing something else.

    ``precision``: 'fp32' (exact-fp32 MFMA, matches the CPU reference to ~1e-6), 'fp32x3' (fp32 operands split into bf16 hi + lo
    inside the GEMMs: the same results to a few 1e-6) or 'bf16' (bf16 MFMA operands with
    fp32 accumulation — the counterpart of the ``torch.cuda.amp.autocast()`` the reference wraps around
    forward_features, utils/feature_extraction.py:35-36)."""

    def __init__(self, volume_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, representation_size=None, distilled=False,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., embed_layer=PatchEmbed3D, norm_layer=None,
                 act_layer=None, weight_init='', global_pool=False, precision=None):
        super().__init__()
        if distilled or representation_size:
            raise NotImplementedError('distilled / representation_size variants are not built for MI355X')
        if act_layer not in (None, nn.GELU):
            raise NotImplementedError('only the exact-GELU MLP of the reference is built')
        if weight_init not in ('', 'nlhb'):
            raise NotImplementedError("only the default ('') and 'nlhb' weight_init schemes are supported")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        self.num_heads = num_heads
        # dropout / stochastic depth are identities in eval mode; kept so the reference's hyper-parameters are accepted
        self.drop_rate, self.attn_drop_rate, self.drop_path_rate = drop_rate, attn_drop_rate, drop_path_rate
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = embed_layer(volume_size=volume_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = None
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.Sequential(*[
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                  attn_drop=attn_drop_rate, norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.ln_eps = self.norm.eps
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.head_dist = None
        self.global_pool = global_pool
        if self.global_pool:
            self.fc_norm = norm_layer(embed_dim)
            del self.norm  # as the reference: fc_norm replaces norm (model/vit.py:218-221)
        self._precision = precision or 'fp32'
        self._encoder = None
        self.init_weights(weight_init)

    def init_weights(self, mode=''):
        head_bias = -math.log(self.num_classes) if 'nlhb' in mode else 0.
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        for name, mod in self.named_modules():
            _init_vit_weights(mod, name, head_bias)

    def _init_weights(self, m):
        _init_vit_weights(m)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=''):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def set_precision(self, precision: str):
        self._precision = precision
        self._encoder = None

    def forward_features(self, x):
        """[B, C, Lz, Hy, Wx] -> [B, embed_dim] (reference model/vit.py:265-284)."""
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            raise VitaeError('VisionTransformer3D on MI355X is inference-only (feature extraction); call it under '
                             'torch.no_grad() / model.eval() — fine-tuning is not built yet')
        if (self.training and (self.drop_rate or self.attn_drop_rate or self.drop_path_rate)):
            raise VitaeError('dropout / stochastic depth are not implemented; use model.eval()')
        if self._encoder is None:
            from ..encoder import HipEncoder
            self._encoder = HipEncoder(self, self._precision)
        return self._encoder.forward_features(x)

    def forward(self, x):
        f = self.forward_features(x)
        if isinstance(self.head, nn.Identity):
            return f
        return hip_linear(f, self.head.weight, self.head.bias)


class VisionTransformer3DContrastive(VisionTransformer3D):
    """Reference model/vit.py:300-340 (SimSiam fine-tuning variant): a training-only model, not part of the
    feature-extraction row; constructing it says so."""

    def __init__(self, *a, **k):
        raise NotImplementedError('VisionTransformer3DContrastive is a training-only down-stream model; the MI355X path '
                                  'covers MAE pre-training and encoder-only feature extraction')
