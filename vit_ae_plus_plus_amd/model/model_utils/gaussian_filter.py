"""Gaussian smoothing of the reconstruction target (reference:
model/model_utils/gaussian_filter.py:5-26), as three separable 11-tap passes on the GPU."""
import torch

from ..._abi import VitaeError, lib
from ...engine import gaussian_taps_host


def make_gaussian_kernel(sigma):
    """fp32 taps with the reference's sampling (linspace(-ks//2, ks//2+1, ks), gaussian_filter.py:9)."""
    return torch.from_numpy(gaussian_taps_host(float(sigma)))


def perform_3d_gaussian_blur(original_vol, blur_sigma=2):
    if not original_vol.is_cuda:
        raise VitaeError('perform_3d_gaussian_blur: MI355X only (no CPU fallback)')
    B, C, L, H, W = original_vol.shape
    x = original_vol.detach().contiguous().float()
    taps = gaussian_taps_host(float(blur_sigma))
    tmp, out = torch.empty_like(x), torch.empty_like(x)
    lib.vitae_gauss_blur_fwd(x.data_ptr(), tmp.data_ptr(), out.data_ptr(), taps.ctypes.data, len(taps), B * C, L, H, W,
                             torch.cuda.current_stream(x.device).cuda_stream)
    return out
