"""Encoder-only inference on MI355X: the kernel sequence behind ``VisionTransformer3D.forward_features``
(reference model/vit.py:265-284, driven by utils/feature_extraction.py:9-45 after pre-training).

Same kernels as the training engine's encoder, unmasked (every patch is kept, N = L + 1 tokens):
patch gather -> patch-embedding GEMM -> cls/pos assembly -> depth x [LN, qkv, attention, proj(+res), LN,
fc1+GELU, fc2(+res)] -> global-pool mean + fc_norm, or norm of the cls rows.  Nothing is kept for a
backward pass, so two activation buffers ping-pong through the blocks.

bf16 mode (the counterpart of the reference's ``torch.cuda.amp.autocast()`` around forward_features)
keeps GEMM operands in bf16 written by their producers and runs every Linear on the LDS-DMA GEMM when
all contraction lengths are multiples of 64; otherwise, and in fp32 mode, the generic Linear launcher
(exact-fp32 MFMA or bf16 MFMA with fp32 activations) is used.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from ._abi import CONSTS as _C, VitaeError, lib

PREC = {'fp32': _C['VITAE_PREC_F32'], 'bf16': _C['VITAE_PREC_BF16'], 'fp32x3': _C['VITAE_PREC_BF16X3']}
EPI_NONE, EPI_GELU = _C['VITAE_EPI_NONE'], _C['VITAE_EPI_GELU']


def _ptr(t):
    return None if t is None else t.data_ptr()


class HipEncoder:
    """Sequences the encoder kernels for one ``VisionTransformer3D`` instance (weights are read from the
    module's parameters at call time; bf16 copies are cached per parameter version)."""

    def __init__(self, module, precision: str = 'fp32'):
        if precision not in PREC:
            raise VitaeError(f'unknown precision {precision!r}')
        lib.load()
        self.m = module
        self.precision, self.prec = precision, PREC[precision]
        self._w16: Dict[str, Tuple[int, torch.Tensor]] = {}
        self._B = None
        self.buf: Dict[str, torch.Tensor] = {}
        self._split: Dict[Tuple, int] = {}

    # ------------------------------------------------------------------ parameters
    def _param(self, name: str):
        p = self.sd.get(name)
        if p is None:        # e.g. qkv_bias=False
            return None
        if p.dtype != torch.float32 or not p.is_contiguous() or p.device != self.device:
            raise VitaeError(f'parameter {name} must be contiguous fp32 on {self.device}')
        return p

    def _bf16(self, name: str) -> int:
        p = self._param(name)
        ent = self._w16.get(name)
        if ent is None or ent[0] != p._version or ent[1].device != p.device:
            t = torch.empty(p.numel(), dtype=torch.bfloat16, device=p.device)
            lib.vitae_cast_bf16(p.data_ptr(), t.data_ptr(), p.numel(), self.stream)
            ent = (p._version, t)
            self._w16[name] = ent
        return ent[1].data_ptr()

    # ------------------------------------------------------------------ workspace
    def _alloc(self, B: int):
        m = self.m
        L, D, H, P = m.patch_embed.num_patches, m.embed_dim, self.hidden, self.P
        N = L + 1
        M = B * N
        if self._B == (B, self.device):
            return
        self._B = (B, self.device)
        dev = self.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        b = self.buf = {}
        b['ids'] = torch.arange(L, dtype=torch.int32, device=dev).repeat(B, 1).contiguous()
        b['tok'] = f(B * L, D)
        b['xa'], b['xb'], b['xmid'] = f(M, D), f(M, D), f(M, D)
        b['qkv'], b['o'], b['lse'] = f(M, 3 * D), f(M, D), f(B * m.num_heads * N)
        b['mean'], b['rstd'] = f(max(M, B)), f(max(M, B))
        b['hpre'] = f(M, H)
        b['pool'], b['feat'] = f(B, D), f(B, D)
        if self.act16:
            z16 = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
            b['patches_16'] = z16(B * L, P)
            b['y_16'], b['o_16'], b['act_16'] = z16(M, D), z16(M, D), z16(M, H)
            b['ws16'] = torch.zeros(1 << 22, dtype=torch.float32, device=dev)
        else:
            b['patches'], b['y'], b['act'] = f(B * L, P), f(M, D), f(M, H)
            b['ws'] = f(1 << 22)

    # ------------------------------------------------------------------ launch helpers
    def _g16(self, x16, wname, bname, M, N, K, y=None, y16=None, epi=EPI_NONE, aux=None, res=None):
        key = ('g', M, N, K, epi)
        s = self._split.get(key)
        if s is None:
            s = 1 if epi == EPI_GELU else lib.vitae_gemm_glds_pick_split_k(M, N, K)
            while s > 1 and lib.vitae_gemm_glds_ws_floats(M, N, s) > self.buf['ws16'].numel():
                s -= 1
            self._split[key] = s
        lib.vitae_gemm_glds(1, 1, _ptr(x16), K, self._bf16(wname), K, _ptr(y), N, _ptr(y16), N, M, N, K,
                            _ptr(self._param(bname)), _ptr(res), N, epi, _ptr(aux), N, 0, s, self.buf['ws16'].data_ptr(), None,
                            self.stream)

    def _lin(self, x, wname, bname, y, M, N, K, epi=EPI_NONE, aux=None, res=None):
        key = ('l', M, N, K, epi)
        s = self._split.get(key)
        if s is None:
            s = 1 if epi != EPI_NONE else lib.vitae_gemm_pick_split_k(M, N, K)
            while s > 1 and s * M * N > self.buf['ws'].numel():
                s -= 1
            self._split[key] = s
        lib.vitae_linear_fwd(self.prec, _ptr(x), _ptr(self._param(wname)), _ptr(self._param(bname)), _ptr(y), M, N, K, epi,
                             _ptr(aux), _ptr(res), s, self.buf['ws'].data_ptr(), self.stream)

    def _ln(self, x, pre, y, y16, M, D):
        b = self.buf
        lib.vitae_layernorm_fwd(_ptr(x), _ptr(self._param(pre + 'weight')), _ptr(self._param(pre + 'bias')), _ptr(y), _ptr(y16),
                                _ptr(b['mean']), _ptr(b['rstd']), M, D, self.eps, self.stream)

    def _sdpa(self, B, N):
        b, m = self.buf, self.m
        if self.prec == PREC['bf16'] and self.hd in (32, 64):
            lib.vitae_sdpa_mfma_fwd(_ptr(b['qkv']), _ptr(b['o']), _ptr(b['o_16']) if self.act16 else None, _ptr(b['lse']), B, N,
                                    m.num_heads, self.hd, self.stream)
        else:
            lib.vitae_sdpa_fwd(_ptr(b['qkv']), _ptr(b['o']), _ptr(b['lse']), B, N, m.num_heads, self.hd, self.stream)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        m = self.m
        if not x.is_cuda:
            raise VitaeError(f'forward_features: input is on {x.device}; this package computes on MI355X only (no CPU fallback)')
        self.device = x.device
        self.stream = torch.cuda.current_stream(x.device).cuda_stream
        self.sd = dict(m.named_parameters())
        pe = m.patch_embed
        ps = pe.patch_size[0]
        if pe.patch_size != (ps, ps, ps) or ps % 4:
            raise VitaeError('patch_size must be cubic and a multiple of 4')
        B, C, Lz, Hy, Wx = x.shape
        assert (Lz, Hy, Wx) == tuple(pe.volume_size), \
            f"Volume image size ({Lz}*{Hy}*{Wx}) doesn't match model ({pe.volume_size[0]}*{pe.volume_size[1]}*{pe.volume_size[2]})."
        L, D = pe.num_patches, m.embed_dim
        N, M = L + 1, B * (L + 1)
        self.P = C * ps ** 3
        self.hidden = m.blocks[0].mlp.fc1.out_features if len(m.blocks) else D
        self.hd = D // m.num_heads
        self.eps = m.ln_eps
        H, P = self.hidden, self.P
        self.act16 = (self.prec == PREC['bf16'] and all(v % 64 == 0 for v in (D, H, P)) and self.hd in (32, 64))
        self._alloc(B)
        b, st = self.buf, self.stream
        xc = x.contiguous().float()
        a16 = self.act16
        lib.vitae_gather_patches(_ptr(xc), _ptr(b['ids']), None if a16 else _ptr(b['patches']), _ptr(b['patches_16']) if a16 else None,
                                 B, C, Lz, Hy, Wx, ps, L, st)
        if a16:
            self._g16(b['patches_16'], 'patch_embed.proj.weight', 'patch_embed.proj.bias', B * L, D, P, y=b['tok'])
        else:
            self._lin(b['patches'], 'patch_embed.proj.weight', 'patch_embed.proj.bias', b['tok'], B * L, D, P)
        lib.vitae_encoder_assemble_fwd(_ptr(b['tok']), _ptr(self._param('cls_token')), _ptr(self._param('pos_embed')),
                                       _ptr(b['ids']), _ptr(b['xa']), B, L, L, D, st)
        cur, nxt = b['xa'], b['xb']
        for i in range(len(m.blocks)):
            q = f'blocks.{i}.'
            if a16:
                self._ln(cur, q + 'norm1.', None, b['y_16'], M, D)
                self._g16(b['y_16'], q + 'attn.qkv.weight', q + 'attn.qkv.bias', M, 3 * D, D, y=b['qkv'])
                self._sdpa(B, N)
                self._g16(b['o_16'], q + 'attn.proj.weight', q + 'attn.proj.bias', M, D, D, y=b['xmid'], res=cur)
                self._ln(b['xmid'], q + 'norm2.', None, b['y_16'], M, D)
                self._g16(b['y_16'], q + 'mlp.fc1.weight', q + 'mlp.fc1.bias', M, H, D, y16=b['act_16'], epi=EPI_GELU, aux=b['hpre'])
                self._g16(b['act_16'], q + 'mlp.fc2.weight', q + 'mlp.fc2.bias', M, D, H, y=nxt, res=b['xmid'])
            else:
                self._ln(cur, q + 'norm1.', b['y'], None, M, D)
                self._lin(b['y'], q + 'attn.qkv.weight', q + 'attn.qkv.bias', b['qkv'], M, 3 * D, D)
                self._sdpa(B, N)
                self._lin(b['o'], q + 'attn.proj.weight', q + 'attn.proj.bias', b['xmid'], M, D, D, res=cur)
                self._ln(b['xmid'], q + 'norm2.', b['y'], None, M, D)
                self._lin(b['y'], q + 'mlp.fc1.weight', q + 'mlp.fc1.bias', b['act'], M, H, D, epi=EPI_GELU, aux=b['hpre'])
                self._lin(b['act'], q + 'mlp.fc2.weight', q + 'mlp.fc2.bias', nxt, M, D, H, res=b['xmid'])
            cur, nxt = nxt, cur
        if m.global_pool:
            lib.vitae_mean_pool_tokens(_ptr(cur), _ptr(b['pool']), B, N, D, 1, st)
            self._ln(b['pool'], 'fc_norm.', b['feat'], None, B, D)
        else:
            # LayerNorm is row-wise: normalising only the cls rows equals norm(x)[:, 0] (model/vit.py:281-282)
            b['pool'].copy_(cur.view(B, N, D)[:, 0])
            self._ln(b['pool'], 'norm.', b['feat'], None, B, D)
        return b['feat'].clone()
