"""ORACLE (test infrastructure): the reference's per-item volume normalisation, restated for a batch.

  dataset/brats_dataset/brats.py:26-37   FlairData._normalize_data / _min_max_normalize_data
  dataset/egd_dataset/egd.py:44-55       the EGD dataset's per-channel z-score variant
Pinned by tests/golden/input_norm.npz, produced by calling the reference's own methods (oracle/gen_golden.py)."""
import torch


def normalize_data(vol: torch.Tensor, use_z_score: bool, per_channel: bool = False) -> torch.Tensor:
    """``vol`` [C, Lz, Hy, Wx] (one item, as the Dataset sees it)."""
    if use_z_score:
        if per_channel:                                                       # egd.py:45-47
            return (vol - vol.mean(dim=[1, 2, 3], keepdim=True)) / torch.sqrt(vol.var(dim=[1, 2, 3], keepdim=True))
        return (vol - vol.mean()) / torch.sqrt(vol.var())                     # brats.py:27-29
    mx, mn = vol.max(), vol.min()                                             # brats.py:30-32
    return 2 * ((vol - mn) / (mx - mn)) - 1


def min_max_normalize_data(vol: torch.Tensor) -> torch.Tensor:                # brats.py:34-37
    mx, mn = vol.max(), vol.min()
    return (vol - mn) / (mx - mn)
