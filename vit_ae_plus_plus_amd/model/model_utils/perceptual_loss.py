"""Perceptual-loss hook (reference: model/model_utils/perceptual_loss.py:11-77), forward only, on the HIP path.

The reference pushes every (batch, z) slice of every channel of the predicted and of the target volume through the first
ten 3x3 convolutions of a VGG16 (``features[0:23]``, split into four slices ending at relu1_2 / relu2_2 / relu3_3 /
relu4_3) and averages the MSE of the four feature maps over slices and channels (:65-77).  It is evaluated under
``torch.no_grad()`` from ``torch.as_tensor`` values (model/vit_autoenc.py:229-230), so it never contributes gradient: a
logging term, weight 0 in the shipped configuration.

What is NOT available here: torchvision (the module tree of ``tv.vgg16``) and the checkpoint ``model/ckp-399.pth`` the
reference loads (:20-24) — neither is part of the reference repository.  The module therefore keeps the reference's
state-dict layout (``slice1.0.weight`` ... ``slice4.21.bias``, so a state dict saved by the reference — or one of
torchvision's ``vgg16().features`` re-keyed — loads directly), starts from torch's default Conv2d initialisation, and its
arithmetic is pinned against a plain ``torch.nn.functional`` restatement (oracle/percep_ref.py) on identical weights.
Parity with torchvision's pretrained VGG16 is UNPINNED.

Kernels: the convolutions are bf16 LDS-DMA GEMMs (``vitae_gemm_glds`` with the bias + ReLU epilogue) over im2col matrices in
NHWC order (``csrc/percep.hip``); fp32 accumulation, bf16 feature maps.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ..._abi import CONSTS, VitaeError, lib

# (slice, index inside vgg16.features, Cin, Cout); a 2x2 max-pool precedes the first convolution of slices 2-4
_VGG = [(1, 0, 3, 64), (1, 2, 64, 64), (2, 5, 64, 128), (2, 7, 128, 128), (3, 10, 128, 256), (3, 12, 256, 256), (3, 14, 256, 256),
        (4, 17, 256, 512), (4, 19, 512, 512), (4, 21, 512, 512)]
_LAST_OF_SLICE = {2: 1, 7: 2, 14: 3, 21: 4}


class vgg_perceptual_loss(nn.Module):
    def __init__(self, requires_grad=False, use_imagenet=False):
        super().__init__()
        if requires_grad:
            raise VitaeError('the perceptual term is forward-only (the reference evaluates it under torch.no_grad())')
        self.use_imagenet = use_imagenet
        self.N_slices = 4
        for s in range(1, 5):
            setattr(self, f'slice{s}', nn.Module())
        for s, idx, cin, cout in _VGG:
            conv = nn.Conv2d(cin, cout, 3, padding=1)
            for p in conv.parameters():
                p.requires_grad = False
            getattr(self, f'slice{s}').add_module(str(idx), conv)
        self._packed = None
        self.chunk_bytes = 2 << 30          # im2col scratch budget per chunk of images
        # The reference fills this stack from model/ckp-399.pth (perceptual_loss.py:20-23) or torchvision's ImageNet weights
        # (use_imagenet) — neither ships with it nor with this image.  Until load_state_dict() has supplied `slice*.*`, the
        # convolutions hold torch's default random initialisation and the term is a number without meaning.
        self._weights_loaded = False
        self._warned = False
        self.register_load_state_dict_post_hook(self._note_loaded)

    @staticmethod
    def _note_loaded(module, incompatible):
        import re
        mine = [k for k in incompatible.missing_keys if re.search(r'(^|\.)slice\d\.\d+\.(weight|bias)$', k)]
        if not mine:                      # only ever SET: a later parent load_state_dict(strict=False) without the VGG keys must
            module._weights_loaded = True     # not turn loaded weights back into "random" 
        module._packed = None

    def convs(self):
        return [(s, idx, getattr(getattr(self, f'slice{s}'), str(idx))) for s, idx, _, _ in _VGG]

    def _pack(self, device):
        """bf16 GEMM operands: W[cout][ky][kx][cin] (the im2col order); the first layer summed over its three identical
        input channels and zero-padded to K = 64."""
        key = (device, tuple(c.weight._version for _, _, c in self.convs()))
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        out = []
        for i, (s, idx, conv) in enumerate(self.convs()):
            w = conv.weight.detach().to(device=device, dtype=torch.float32)
            if i == 0:
                w9 = w.sum(1).reshape(w.shape[0], 9)
                wk = torch.zeros(w.shape[0], 64, device=device)
                wk[:, :9] = w9
            else:
                wk = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
            out.append((wk.to(torch.bfloat16).contiguous(), conv.bias.detach().to(device=device, dtype=torch.float32).contiguous()))
        self._packed = (key, out)
        return out

    def _warn_if_random(self):
        if not self._weights_loaded and not self._warned:
            import warnings
            self._warned = True
            warnings.warn('vgg_perceptual_loss: no VGG16 weights have been loaded (the reference reads model/ckp-399.pth'
                          + (' / torchvision ImageNet weights, use_imagenet=True' if self.use_imagenet else '') +
                          '): the perceptual term is computed with RANDOM convolution weights and is folded into the reported '
                          'loss as is; load a reference-format state dict (keys perceptual_loss.slice*.*) first.  Note that a '
                          'non-zero perceptual_weight also takes the model off the fused (captured) step: the generic route runs.',
                          RuntimeWarning, stacklevel=3)

    @torch.no_grad()
    def forward(self, X1, X2):
        """mean over channels of mean over the four slices of MSE(features(X1 slices), features(X2 slices)); X: [B, C, Z, H, W].
        Evaluated under torch.no_grad() like the reference's logging term (model/vit_autoenc.py:229-230)."""
        self._warn_if_random()
        if not X1.is_cuda:
            raise VitaeError('vgg_perceptual_loss: MI355X only (no CPU fallback; the CPU restatement is oracle/percep_ref.py)')
        X1, X2 = X1.contiguous().float(), X2.contiguous().float()
        B, C, Z, H, W = X1.shape
        if H % 8 or W % 8:
            raise VitaeError('perceptual loss: slice height / width must be multiples of 8 (three 2x2 poolings)')
        dev = X1.device
        packed = self._pack(dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        RELU = CONSTS['VITAE_EPI_RELU']
        n_all = B * Z
        per_img = H * W * 9 * 64 * 2 * 2          # largest im2col (conv1_2), both volumes, bytes per image
        n_chunk = max(1, min(n_all, self.chunk_bytes // per_img))
        acc = torch.zeros(C, 4, dtype=torch.float64, device=dev)
        counts = [0.0] * 4
        bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
        for ch in range(C):
            for img0 in range(0, n_all, n_chunk):
                n = min(n_chunk, n_all - img0)
                n2, h, w = 2 * n, H, W
                A = bf(n2 * h * w, 64)
                lib.vitae_percep_im2col_first(X1.data_ptr(), X2.data_ptr(), A.data_ptr(), B, C, ch, Z, H, W, img0, n, st)
                feat = None
                for i, ((wk, bias), (s, idx, cin, cout)) in enumerate(zip(packed, _VGG)):
                    if i > 0:
                        if idx in (5, 10, 17):      # the max-pool that opens slices 2-4
                            pooled = bf(n2 * (h // 2) * (w // 2), cin)
                            lib.vitae_percep_maxpool2(feat.data_ptr(), pooled.data_ptr(), n2, h, w, cin, st)
                            feat, h, w = pooled, h // 2, w // 2
                        A = bf(n2 * h * w, 9 * cin)
                        lib.vitae_percep_im2col(feat.data_ptr(), A.data_ptr(), n2, h, w, cin, st)
                    M, K = A.shape
                    out = bf(M, cout)
                    lib.vitae_gemm_glds(1, 1, A.data_ptr(), K, wk.data_ptr(), K, None, cout, out.data_ptr(), cout, M, cout, K,
                                        bias.data_ptr(), None, 0, RELU, None, 0, 0, 1, None, None, st)
                    feat = out
                    sl = _LAST_OF_SLICE.get(idx)
                    if sl is not None:
                        half = (M // 2) * cout
                        lib.vitae_percep_sqdiff(feat.data_ptr(), feat.data_ptr() + 2 * half, half, acc[ch, sl - 1:].data_ptr(), st)
                        if ch == 0:
                            counts[sl - 1] += half
                    del A
        mse = acc / torch.tensor(counts, dtype=torch.float64, device=dev)
        return mse.mean(1).mean().float()
