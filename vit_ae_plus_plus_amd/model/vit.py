"""Building blocks of the 3-D ViT with the reference's names and parameter layout
(reference: model/vit.py:52-144).

In this package the modules are *parameter containers*: they own tensors under the reference's
state-dict keys (``proj.weight``, ``attn.qkv.weight``, ``mlp.fc1.bias`` ...) so checkpoints
round-trip, while the arithmetic of a whole model is sequenced by ``HipMAEEngine`` over
libvitae_hip.so.  Their stand-alone ``forward`` methods run the same HIP kernels op by op
(inference only: used by feature extraction, not by the training hot path).
"""
from __future__ import annotations

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


class VisionTransformer3D(nn.Module):
    """Down-stream encoder-only ViT (reference model/vit.py:147-298).  Listed under SURVEY §8(f)
    'next rows' (feature extraction); not part of the round-1 training hot path."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError('VisionTransformer3D (feature extraction / fine-tuning) is a §8(f) "next" row; '
                                  'the MI355X path currently covers MAE pre-training')


class VisionTransformer3DContrastive(VisionTransformer3D):
    pass
