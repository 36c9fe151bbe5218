"""ORACLE (test infrastructure, not product code): CPU restatement of the augmentations the reference's pre-training
scripts put in front of the hot path:

  k_fold_training_scripts/k_fold_cross_valid_combined_brats.py:93-97
      transforms = [tio.RandomAffine(), tio.RandomNoise(std=0.1), tio.RandomGamma(log_gamma=(-0.3, 0.3))]
  dataset/brats_dataset/brats.py:39-44
      original_volume = normalize(volume);  volume = normalize(transform(volume))        (per item, raw [C, L, H, W] tensor)

PARITY UNPINNED.  The arithmetic lives in third-party code that is not under /root/reference and not installed here:
torchio==0.18.73 and SimpleITK==2.2.1 (requirements.txt:12,16).  Their published behaviour is restated below; what CAN
be pinned here is pinned in tests/test_augment.py: the interpolation arithmetic against scipy.ndimage.affine_transform
(order=1), an independent implementation, and exact identities (identity matrix, integer shifts, gamma = 1, std = 0).

Restated conventions (torchio.transforms.augmentation.spatial.random_affine / intensity.random_noise / random_gamma):
  * RandomAffine(): scales=0.1 -> three factors ~ U(0.9, 1.1); degrees=10 -> three angles ~ U(-10, 10) degrees;
    translation=0; center='image'; default_pad_value='minimum'; linear interpolation.  A bare tensor is an image with
    identity affine, so physical coordinates are voxel indices up to torchio's RAS->LPS sign flips, which cancel against
    the flips it applies to the angles: in index space the resampling reads
        src = c + S R (dst - c) + t,      c = (n - 1) / 2,   S = diag(scales),   R = Rz(rz) Rx(rx) Ry(ry)
    (SimpleITK's Euler3DTransform default ZXY order; CompositeTransform([scale, rotate]) applies the rotation first),
    axis order = the tensor's own (l, h, w).  sitk.Resample writes the default value where src falls outside
    [-0.5, n - 0.5) on any axis and interpolates linearly elsewhere, clamping neighbours at the half-voxel border.
  * RandomNoise(std=0.1): std ~ U(0, 0.1), x + N(0, std), one std per item, noise for every element.
  * RandomGamma(log_gamma=(-0.3, 0.3)): gamma = exp(U(-0.3, 0.3)); x ** gamma, with sign(x) |x| ** gamma when the item
    has negative values (torchio's documented behaviour for such images).

Only ``tests/`` may import this module.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def rotation_zxy(deg) -> np.ndarray:
    """R = Rz(rz) Rx(rx) Ry(ry), angles in degrees about axes 0, 1, 2 of the tensor."""
    rx, ry, rz = (math.radians(float(v)) for v in deg)
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float64)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float64)
    return Rz @ Rx @ Ry


def affine_matrix(scales, degrees, translation, shape) -> np.ndarray:
    """3x4 matrix A with src = A (dst, 1): src = c + S R (dst - c) + t."""
    c = (np.asarray(shape, dtype=np.float64) - 1.0) / 2.0
    M = np.diag(np.asarray(scales, dtype=np.float64)) @ rotation_zxy(degrees)
    off = c - M @ c + np.asarray(translation, dtype=np.float64)
    return np.concatenate([M, off[:, None]], axis=1)


def affine_resample(vol: torch.Tensor, A: np.ndarray, pad: float) -> torch.Tensor:
    """``vol`` [C, L, H, W] float32; linear interpolation at src = A (dst, 1), float32 arithmetic like the kernel."""
    C, L, H, W = vol.shape
    A = torch.as_tensor(A, dtype=torch.float32)
    g = torch.stack(torch.meshgrid(torch.arange(L, dtype=torch.float32), torch.arange(H, dtype=torch.float32),
                                   torch.arange(W, dtype=torch.float32), indexing='ij'), dim=0)      # [3, L, H, W]
    s = [A[d, 0] * g[0] + A[d, 1] * g[1] + A[d, 2] * g[2] + A[d, 3] for d in range(3)]
    n = (L, H, W)
    inside = torch.ones(L, H, W, dtype=torch.bool)
    for d in range(3):
        inside &= (s[d] >= -0.5) & (s[d] < n[d] - 0.5)
    base = [torch.floor(v) for v in s]
    t = [s[d] - base[d] for d in range(3)]
    i0 = [base[d].long().clamp(0, n[d] - 1) for d in range(3)]
    i1 = [(base[d].long() + 1).clamp(0, n[d] - 1) for d in range(3)]

    def at(il, ih, iw):
        return vol[:, il, ih, iw]

    def lerp(a, b, w):
        return a + w * (b - a)
    a00 = lerp(at(i0[0], i0[1], i0[2]), at(i0[0], i0[1], i1[2]), t[2])
    a01 = lerp(at(i0[0], i1[1], i0[2]), at(i0[0], i1[1], i1[2]), t[2])
    a10 = lerp(at(i1[0], i0[1], i0[2]), at(i1[0], i0[1], i1[2]), t[2])
    a11 = lerp(at(i1[0], i1[1], i0[2]), at(i1[0], i1[1], i1[2]), t[2])
    out = lerp(lerp(a00, a01, t[1]), lerp(a10, a11, t[1]), t[0])
    return torch.where(inside.unsqueeze(0), out, torch.full_like(out, float(pad)))


def random_affine(vol: torch.Tensor, scales, degrees, translation=(0, 0, 0), pad='minimum') -> torch.Tensor:
    A = affine_matrix(scales, degrees, translation, vol.shape[1:])
    return affine_resample(vol, A, float(vol.min()) if pad == 'minimum' else float(pad))


def random_noise(vol: torch.Tensor, std: float, noise: torch.Tensor) -> torch.Tensor:
    return vol + float(std) * noise


def random_gamma(vol: torch.Tensor, gamma: float) -> torch.Tensor:
    if gamma == 1.0:
        return vol.clone()
    return torch.sign(vol) * vol.abs() ** float(gamma)
