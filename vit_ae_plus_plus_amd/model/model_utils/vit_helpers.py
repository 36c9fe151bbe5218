"""Fixed 3-D sin-cos position tables (reference: model/model_utils/vit_helpers.py:13-70).

Host-side, float64 numpy, computed once per model; generalised to non-cubic grids.
"""
import numpy as np
import torch


def _axis_table(width: int, coord: np.ndarray) -> np.ndarray:
    """[sin(coord * w_k) | cos(coord * w_k)], w_k = 10000^(-k / (width/2))  (vit_helpers.py:48-70)."""
    if width % 2:
        raise ValueError('sin-cos width must be even')
    half = width // 2
    freq = np.power(10000.0, -np.arange(half, dtype=np.float64) / half)
    ang = coord.astype(np.float64).reshape(-1, 1) * freq.reshape(1, -1)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def get_3d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """Same values as the reference function for an int ``grid_size``; also accepts a (gl, gh, gw)
    tuple.  Token (l, h, w) gets the column blocks [enc(h) | enc(l) | enc(w)]: the reference builds
    its coordinate grid with np.meshgrid's default 'xy' indexing (vit_helpers.py:22), which swaps the
    first two axes.  Widths: D//3 rounded up to even twice, remainder for w
    (vit_helpers.py:36-42).  Returns float64 [(1+) gl*gh*gw, embed_dim]."""
    gl, gh, gw = (grid_size,) * 3 if np.isscalar(grid_size) else tuple(int(g) for g in grid_size)
    if embed_dim % 2:
        raise ValueError('embed_dim must be even')
    third = embed_dim // 3
    third += third % 2
    zl, zh, zw = np.indices((gl, gh, gw))
    table = np.concatenate([_axis_table(third, zh), _axis_table(third, zl),
                            _axis_table(embed_dim - 2 * third, zw)], axis=1)
    if cls_token:
        table = np.concatenate([np.zeros((1, embed_dim)), table], axis=0)
    return table


def interpolate_pos_embed(model, checkpoint_model):
    """Resize a checkpoint's ``pos_embed`` to the model's patch grid (vit_helpers.py:180-204):
    extra (cls) tokens kept, patch tokens tri-linearly interpolated on the cubic grid."""
    if 'pos_embed' not in checkpoint_model:
        return
    ckpt = checkpoint_model['pos_embed']
    dim = ckpt.shape[-1]
    n_patches = model.patch_embed.num_patches
    n_extra = model.pos_embed.shape[-2] - n_patches
    old = round((ckpt.shape[-2] - n_extra) ** (1 / 3))
    new = round(n_patches ** (1 / 3))
    if old == new:
        return
    print('Position interpolate from %dx%dx%d to %dx%dx%d' % (old, old, old, new, new, new))
    extra, grid = ckpt[:, :n_extra], ckpt[:, n_extra:]
    grid = grid.reshape(-1, old, old, old, dim).permute(0, 4, 1, 2, 3)
    grid = torch.nn.functional.interpolate(grid, size=(new, new, new), mode='trilinear', align_corners=False)
    grid = grid.permute(0, 2, 3, 4, 1).flatten(1, 3)
    checkpoint_model['pos_embed'] = torch.cat((extra, grid), dim=1)
