"""SURVEY §8(f) row 4 — the torchio augmentations of the pre-training scripts as batch GPU ops
(k_fold_cross_valid_combined_brats.py:93-97, applied in dataset/brats_dataset/brats.py:39-44).

torchio / SimpleITK are absent, so parity with them is UNPINNED (oracle/augment_ref.py lists the restated conventions).
What is pinned: the oracle's interpolation against scipy.ndimage.affine_transform (an independent implementation),
exact identities, and the HIP kernels against the oracle on identical parameters and noise."""
import numpy as np
import pytest
import torch

from oracle import augment_ref as A


def _vol(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * 3 + 1


def test_oracle_interpolation_matches_scipy():
    from scipy import ndimage
    vol = _vol((2, 13, 11, 9), 1)
    M = A.affine_matrix((1.07, 0.93, 1.02), (7.0, -4.0, 9.5), (0.3, -0.6, 0.2), vol.shape[1:])
    got = A.affine_resample(vol, M, pad=-123.0).numpy()
    g = np.stack(np.meshgrid(*[np.arange(n, dtype=np.float64) for n in vol.shape[1:]], indexing='ij'), 0).reshape(3, -1)
    src = (M[:, :3] @ g + M[:, 3:]).reshape(3, *vol.shape[1:])
    interior = np.ones(vol.shape[1:], bool)
    for d, n in enumerate(vol.shape[1:]):
        interior &= (src[d] >= 0) & (src[d] <= n - 1)
    assert interior.mean() > 0.4
    for c in range(vol.shape[0]):
        ref = ndimage.affine_transform(vol[c].double().numpy(), M[:, :3], offset=M[:, 3], order=1, mode='constant', cval=-123.0)
        assert np.abs(got[c] - ref)[interior].max() < 2e-5
    outside = np.zeros(vol.shape[1:], bool)
    for d, n in enumerate(vol.shape[1:]):
        outside |= (src[d] < -0.5) | (src[d] >= n - 0.5)
    assert outside.any() and np.all(got[:, outside] == -123.0)


def test_oracle_identities():
    vol = _vol((3, 8, 7, 6), 2)
    ident = A.affine_matrix((1, 1, 1), (0, 0, 0), (0, 0, 0), vol.shape[1:])
    assert torch.equal(A.affine_resample(vol, ident, 0.0), vol)
    shift = A.affine_matrix((1, 1, 1), (0, 0, 0), (2, 0, -1), vol.shape[1:])      # src = dst + (2, 0, -1)
    out = A.affine_resample(vol, shift, -9.0)
    assert torch.equal(out[:, :6, :, 1:], vol[:, 2:, :, :5])
    assert torch.all(out[:, 6:] == -9.0) and torch.all(out[:, :, :, 0] == -9.0)
    R = A.rotation_zxy((10, -20, 30))
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
    assert torch.equal(A.random_gamma(vol, 1.0), vol)
    assert torch.allclose(A.random_gamma(vol, 2.0), torch.sign(vol) * vol * vol)


def test_host_matrices_match_the_oracle_and_parameter_ranges():
    from vit_ae_plus_plus_amd.utils.augment import RandomAffine, RandomGamma, RandomNoise
    ra = RandomAffine(generator=torch.Generator().manual_seed(3))
    s, d, t = ra.get_params(64)
    assert 0.9 <= float(s.min()) and float(s.max()) <= 1.1 and float(d.abs().max()) <= 10 and float(t.abs().max()) == 0
    m = RandomAffine.matrices(s, d, t, (12, 10, 8))
    for b in (0, 17, 63):
        assert np.allclose(m[b].view(3, 4).numpy(), A.affine_matrix(s[b], d[b], t[b], (12, 10, 8)), atol=1e-6)
    std = RandomNoise(std=0.1, generator=torch.Generator().manual_seed(4)).get_params(256)
    assert 0 <= float(std.min()) and float(std.max()) <= 0.1
    lg = torch.log(RandomGamma(log_gamma=(-0.3, 0.3), generator=torch.Generator().manual_seed(5)).get_params(256))
    assert -0.3 <= float(lg.min()) and float(lg.max()) <= 0.3
    iso = RandomAffine(isotropic=True, generator=torch.Generator().manual_seed(6)).get_params(4)[0]
    assert torch.equal(iso[:, 0], iso[:, 1]) and torch.equal(iso[:, 0], iso[:, 2])


def test_cpu_input_fails_loudly():
    from vit_ae_plus_plus_amd._abi import VitaeError
    from vit_ae_plus_plus_amd.utils.augment import RandomAffine, RandomGamma
    with pytest.raises(VitaeError):
        RandomAffine()(torch.zeros(1, 1, 4, 4, 4))
    with pytest.raises(VitaeError):
        RandomGamma()(torch.zeros(1, 1, 4, 4, 4))


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,shape', [(3, 2, (12, 10, 9)), (2, 4, (32, 32, 32)), (1, 1, (24, 16, 40)), (2, 1, (5, 7, 3))])
def test_hip_augmentations_match_the_oracle(B, C, shape):
    from vit_ae_plus_plus_amd.utils.augment import Compose, RandomAffine, RandomGamma, RandomNoise
    x = _vol((B, C, *shape), 10)
    xd = x.cuda()
    ra = RandomAffine(translation=1.5, generator=torch.Generator().manual_seed(11))
    y = ra(xd).cpu()
    p = ra.last_params
    for b in range(B):
        ref = A.random_affine(x[b], p['scales'][b], p['degrees'][b], p['translation'][b])
        assert float((y[b] - ref).abs().max()) < 2e-4, b           # float32 coordinates on both sides, FMA contraction differs
    # exact identities
    ident = RandomAffine.matrices(torch.ones(B, 3), torch.zeros(B, 3), torch.zeros(B, 3), shape)
    assert torch.equal(ra.apply(xd, ident).cpu(), x)
    fixed = RandomAffine(default_pad_value=-7.0)
    sh = RandomAffine.matrices(torch.ones(B, 3), torch.zeros(B, 3), torch.tensor([[2.0, 0.0, -1.0]] * B), shape)
    out = fixed.apply(xd, sh).cpu()
    assert torch.equal(out[:, :, :shape[0] - 2, :, 1:], x[:, :, 2:, :, :shape[2] - 1]) and torch.all(out[:, :, shape[0] - 2:] == -7.0)
    # noise + gamma, separately and fused in Compose, on the same noise tensor
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(12))
    rn, rg = RandomNoise(std=0.1, generator=torch.Generator().manual_seed(13)), RandomGamma(generator=torch.Generator().manual_seed(14))
    z1 = rg(rn(xd, noise=noise.cuda())).cpu()
    for b in range(B):
        ref = A.random_gamma(A.random_noise(x[b], rn.last_params['std'][b], noise[b]), float(rg.last_params['gamma'][b]))
        assert float((z1[b] - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    rn2, rg2 = RandomNoise(std=0.1, generator=torch.Generator().manual_seed(13)), RandomGamma(generator=torch.Generator().manual_seed(14))
    rn2.draw = lambda t: noise.cuda()
    assert torch.equal(Compose([rn2, rg2])(xd).cpu(), z1)
    assert torch.equal(RandomGamma(log_gamma=0)(xd).cpu(), x)      # gamma = 1 is an exact pass-through


@pytest.mark.gpu
def test_views_of_a_bench_size_batch():
    """Full-size batch through the whole chain: both views come out normalised; an identity chain gives identical views."""
    from vit_ae_plus_plus_amd.utils.augment import Compose, RandomAffine, RandomGamma, RandomNoise, augmented_views
    g = torch.Generator(device='cuda').manual_seed(21)
    ax = torch.arange(96, device='cuda', dtype=torch.float32)
    smooth = torch.sin(ax / 7)[:, None, None] + torch.cos(ax / 9)[None, :, None] * torch.sin(ax / 5)[None, None, :]   # "anatomy"
    raw = 100 + 40 * smooth[None, None] * torch.tensor([1.0, 0.7, 1.3, 0.9], device='cuda').view(1, 4, 1, 1, 1) \
        + torch.randn(2, 4, 96, 96, 96, device='cuda', generator=g)
    tf = Compose([RandomAffine(generator=torch.Generator().manual_seed(22)), RandomNoise(std=0.1, generator=torch.Generator().manual_seed(23)),
                  RandomGamma(log_gamma=(-0.3, 0.3), generator=torch.Generator().manual_seed(24))])
    v1, v2 = augmented_views(raw, tf, use_z_score=True)
    for v in (v1, v2):
        assert abs(float(v.double().mean(dim=(1, 2, 3, 4)).abs().max())) < 1e-4
        assert abs(float(v.double().var(dim=(1, 2, 3, 4)).max()) - 1) < 1e-4
    assert float((v1 - v2).abs().mean()) > 1e-2                     # the views differ ...
    cc = float((v1 * v2).double().mean())
    assert cc > 0.5                                                  # ... but are the same anatomy (small affine, small noise)
    i1, i2 = augmented_views(raw, Compose([RandomAffine(scales=0, degrees=0), RandomGamma(log_gamma=0)]), use_z_score=False)
    assert torch.equal(i1, i2)
