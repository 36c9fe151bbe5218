"""3-D Sobel edge magnitude (reference: model/model_utils/sobel_filter.py:5-45).

``SobelFilter3d`` keeps the frozen ``sobel_filter.{weight,bias}`` tensors of the reference so
state dicts match; the arithmetic is ``vitae_sobel_edge_fwd`` (the three stencils are fixed in the
kernel; the reference never trains them, sobel_filter.py:33-35).
"""
import torch
from torch import nn

from ..._abi import VitaeError, lib


def _sobel_weight():
    s = torch.tensor([1., 2., 1.])
    d = torch.tensor([1., 0., -1.])
    return torch.stack([torch.einsum('i,j,k->ijk', s, s, d), torch.einsum('i,j,k->ijk', s, -d, s),
                        torch.einsum('i,j,k->ijk', -d, s, s)])[:, None]


class SobelFilter3d(nn.Module):
    def __init__(self):
        super().__init__()
        self.sobel_filter = nn.Conv3d(1, 3, kernel_size=3, stride=1, padding=1)
        with torch.no_grad():
            self.sobel_filter.weight.copy_(_sobel_weight())
            self.sobel_filter.bias.zero_()
        for p in self.sobel_filter.parameters():
            p.requires_grad = False

    def forward(self, x):
        """x [B, C, L, H, W] -> [B, L, H, W]: sum over channels of sqrt(gx^2 + gy^2 + gz^2).
        Inference helper (the training path fuses this into the loss chain with its backward)."""
        if not x.is_cuda:
            raise VitaeError('SobelFilter3d: MI355X only (no CPU fallback)')
        B, C, L, H, W = x.shape
        xc = x.detach().contiguous().float()
        out = torch.empty(B, L, H, W, dtype=torch.float32, device=x.device)
        lib.vitae_sobel_edge_fwd(xc.data_ptr(), out.data_ptr(), None, None, B, C, L, H, W,
                                 torch.cuda.current_stream(x.device).cuda_stream)
        return out
