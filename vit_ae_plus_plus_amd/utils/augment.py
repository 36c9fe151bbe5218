"""The augmentations of the pre-training scripts as batch operations on volumes already in HBM.

The reference augments one item at a time in DataLoader workers with torchio
(k_fold_training_scripts/k_fold_cross_valid_combined_brats.py:93-97: ``tio.RandomAffine()``, ``tio.RandomNoise(std=0.1)``,
``tio.RandomGamma(log_gamma=(-0.3, 0.3))``, composed, applied in dataset/brats_dataset/brats.py:39-44) and ships 14 MB
of fp32 per volume and view over PCIe afterwards.  Here the raw batch is uploaded once and both views are produced on
the GPU: the classes below keep torchio's names, constructor arguments and parameter distributions, draw the random
parameters on the host (one set per item, ``torch`` CPU generator) and launch deterministic HIP kernels
(``csrc/input.hip``) for the whole batch.

torchio / SimpleITK are third-party code that is absent here: the conventions restated (and what is and is not pinned)
are listed in ``oracle/augment_ref.py``.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

from .._abi import VitaeError, lib
from .input_pipeline import normalize_data


def _check(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise VitaeError(f'augmentation input is on {x.device}; this package computes on MI355X only (no CPU fallback)')
    if x.dim() != 5:
        raise VitaeError('augmentations take a batch [B, C, Lz, Hy, Wx]')
    return x.contiguous().float()


def _range(v, lo_of_scalar, name) -> Tuple[float, float]:
    if isinstance(v, (int, float)):
        return lo_of_scalar(float(v))
    a, b = (float(t) for t in v)
    if a > b:
        raise ValueError(f'{name}: lower bound {a} above upper bound {b}')
    return a, b


def _stream(x):
    return torch.cuda.current_stream(x.device).cuda_stream


class _Random:
    def __init__(self, generator: Optional[torch.Generator] = None):
        self.generator = generator
        self.last_params = None          # what the last call drew (tests, logging)

    def _uniform(self, n: int, lo: float, hi: float) -> torch.Tensor:
        return torch.empty(n, dtype=torch.float32).uniform_(lo, hi, generator=self.generator)


class RandomAffine(_Random):
    """``tio.RandomAffine``: per item three scale factors ~ U(scales), three Euler angles ~ U(degrees) (degrees, about
    the image centre), a translation ~ U(translation) in voxels; linear resampling; pad value = the item's minimum."""

    def __init__(self, scales=0.1, degrees=10, translation=0, isotropic: bool = False, default_pad_value='minimum',
                 generator: Optional[torch.Generator] = None):
        super().__init__(generator)
        self.scales = _range(scales, lambda d: (1 - d, 1 + d), 'scales')
        self.degrees = _range(degrees, lambda d: (-d, d), 'degrees')
        self.translation = _range(translation, lambda d: (-d, d), 'translation')
        if self.scales[0] <= 0:
            raise ValueError('scales must be positive')
        self.isotropic = isotropic
        if default_pad_value != 'minimum' and not isinstance(default_pad_value, (int, float)):
            raise VitaeError("default_pad_value: 'minimum' or a number ('mean' / 'otsu' are not rebuilt)")
        self.default_pad_value = default_pad_value

    def get_params(self, B: int):
        s = self._uniform(3 * B, *self.scales).view(B, 3)
        if self.isotropic:
            s = s[:, :1].expand(B, 3).contiguous()
        d = self._uniform(3 * B, *self.degrees).view(B, 3)
        t = self._uniform(3 * B, *self.translation).view(B, 3)
        return s, d, t

    @staticmethod
    def matrices(scales: torch.Tensor, degrees: torch.Tensor, translation: torch.Tensor, shape: Sequence[int]) -> torch.Tensor:
        """[B, 12]: row-major 3x4 A_b with  src = c + S R (dst - c) + t,  R = Rz Rx Ry  (float64 on the host)."""
        B = scales.shape[0]
        c = (torch.tensor(shape, dtype=torch.float64) - 1) / 2
        out = torch.empty(B, 3, 4, dtype=torch.float64)
        for b in range(B):
            rx, ry, rz = (math.radians(float(v)) for v in degrees[b])
            cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
            Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float64)
            Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)
            Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float64)
            M = torch.diag(scales[b].double()) @ (Rz @ Rx @ Ry)
            out[b, :, :3] = M
            out[b, :, 3] = c - M @ c + translation[b].double()
        return out.view(B, 12).float()

    def apply(self, x: torch.Tensor, mats: torch.Tensor) -> torch.Tensor:
        x = _check(x)
        B, C, Lz, Hy, Wx = x.shape
        y = torch.empty_like(x)
        m = mats.to(x.device, torch.float32).contiguous()
        if self.default_pad_value == 'minimum':
            ws = torch.empty(3 * B, dtype=torch.float64, device=x.device)
            lib.vitae_volume_minmax(x.data_ptr(), ws.data_ptr(), B, x[0].numel(), _stream(x))
            lib.vitae_affine_resample(x.data_ptr(), y.data_ptr(), m.data_ptr(), ws.data_ptr(), 0.0, B, C, Lz, Hy, Wx, _stream(x))
        else:
            lib.vitae_affine_resample(x.data_ptr(), y.data_ptr(), m.data_ptr(), None, float(self.default_pad_value), B, C, Lz,
                                      Hy, Wx, _stream(x))
        return y

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        s, d, t = self.get_params(x.shape[0])
        self.last_params = {'scales': s, 'degrees': d, 'translation': t}
        return self.apply(x, self.matrices(s, d, t, x.shape[2:]))


def _noise_gamma(x, noise, stds, gammas):
    x = _check(x)
    B = x.shape[0]
    y = torch.empty_like(x)
    dev = x.device
    sd = None if stds is None else stds.to(dev, torch.float32).contiguous()
    gm = None if gammas is None else gammas.to(dev, torch.float32).contiguous()
    lib.vitae_noise_gamma(x.data_ptr(), None if noise is None else noise.data_ptr(), y.data_ptr(),
                          None if sd is None else sd.data_ptr(), None if gm is None else gm.data_ptr(), B, x[0].numel(),
                          _stream(x))
    return y


class RandomNoise(_Random):
    """``tio.RandomNoise``: x + N(mean, std) with one std ~ U(std range) per item (scalar ``std`` d means (0, d))."""

    def __init__(self, mean=0, std=(0, 0.25), generator: Optional[torch.Generator] = None,
                 device_generator: Optional[torch.Generator] = None):
        super().__init__(generator)
        if mean != 0:
            raise VitaeError('RandomNoise: only mean = 0 is rebuilt (the reference uses the default)')
        self.std = _range(std, lambda d: (0.0, d), 'std')
        self.device_generator = device_generator

    def get_params(self, B: int) -> torch.Tensor:
        return self._uniform(B, *self.std)

    def draw(self, x: torch.Tensor) -> torch.Tensor:
        return torch.randn(x.shape, dtype=torch.float32, device=x.device, generator=self.device_generator)

    def __call__(self, x: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        stds = self.get_params(x.shape[0])
        self.last_params = {'std': stds}
        return _noise_gamma(x, self.draw(x) if noise is None else noise, stds, None)


class RandomGamma(_Random):
    """``tio.RandomGamma``: sign(x) |x| ** gamma, gamma = exp(U(log_gamma)) per item (scalar d means (-d, d))."""

    def __init__(self, log_gamma=(-0.3, 0.3), generator: Optional[torch.Generator] = None):
        super().__init__(generator)
        self.log_gamma = _range(log_gamma, lambda d: (-d, d), 'log_gamma')

    def get_params(self, B: int) -> torch.Tensor:
        return torch.exp(self._uniform(B, *self.log_gamma))

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        g = self.get_params(x.shape[0])
        self.last_params = {'gamma': g}
        return _noise_gamma(x, None, None, g)


class Compose:
    """``tio.Compose`` for the classes above.  A RandomNoise directly followed by a RandomGamma runs as one kernel."""

    def __init__(self, transforms: List):
        self.transforms = list(transforms)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        i, t = 0, self.transforms
        while i < len(t):
            if isinstance(t[i], RandomNoise) and i + 1 < len(t) and isinstance(t[i + 1], RandomGamma):
                stds, gam = t[i].get_params(x.shape[0]), t[i + 1].get_params(x.shape[0])
                t[i].last_params, t[i + 1].last_params = {'std': stds}, {'gamma': gam}
                x = _noise_gamma(x, t[i].draw(_check(x)), stds, gam)
                i += 2
            else:
                x = t[i](x)
                i += 1
        return x


def augmented_views(volumes: torch.Tensor, transform=None, use_z_score: bool = True, per_channel: bool = False):
    """What ``FlairData.__getitem__`` returns for each item of a raw batch (brats.py:39-44):
    (normalize(transform(volume)), normalize(volume)) = (view 1, view 2) of the contrastive step."""
    x = _check(volumes)
    original = normalize_data(x, use_z_score, per_channel)
    view1 = normalize_data(transform(x) if transform is not None else x, use_z_score, per_channel)
    return view1, original
