"""Volume normalisation on the GPU: what the reference's Dataset objects do per item on the CPU
(dataset/brats_dataset/brats.py:26-37, dataset/egd_dataset/egd.py:44-55), for a whole batch already in HBM.

The torchio augmentations of the training scripts (k_fold_cross_valid_combined_brats.py:93-97) live in utils/augment.py
(third-party behaviour restated, parity unpinned); the normalisation here is what the reference defines itself and is
pinned against its own methods."""
import torch

from .._abi import CONSTS, VitaeError, lib


def _run(x: torch.Tensor, groups: int, n: int, mode: int, out=None) -> torch.Tensor:
    if not x.is_cuda:
        raise VitaeError(f'normalisation input is on {x.device}; this package computes on MI355X only (no CPU fallback)')
    xc = x.contiguous().float()
    y = torch.empty_like(xc) if out is None else out
    ws = torch.empty(3 * groups, dtype=torch.float64, device=x.device)
    lib.vitae_normalize_volumes(xc.data_ptr(), y.data_ptr(), ws.data_ptr(), groups, n, mode,
                                torch.cuda.current_stream(x.device).cuda_stream)
    return y


def normalize_data(volumes: torch.Tensor, use_z_score: bool = True, per_channel: bool = False) -> torch.Tensor:
    """``volumes`` [B, C, Lz, Hy, Wx] -> each sample normalised like ``Dataset._normalize_data``:
    z-score over the whole sample (BraTS, brats.py:27-29) or per channel (EGD, egd.py:45-47), unbiased variance;
    otherwise min-max of the whole sample to [-1, 1] (brats.py:30-32)."""
    B, C = volumes.shape[:2]
    n = volumes[0, 0].numel()
    if use_z_score:
        return _run(volumes, B * C if per_channel else B, n if per_channel else C * n, CONSTS['VITAE_NORM_ZSCORE'])
    return _run(volumes, B, C * n, CONSTS['VITAE_NORM_MINMAX_PM1'])


def min_max_normalize_data(volumes: torch.Tensor) -> torch.Tensor:
    """Whole-sample min-max to [0, 1] (brats.py:34-37, egd.py:52-55)."""
    B = volumes.shape[0]
    return _run(volumes, B, volumes[0].numel(), CONSTS['VITAE_NORM_MINMAX_01'])
