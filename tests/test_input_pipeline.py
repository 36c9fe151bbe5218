"""SURVEY §8(f) row 4 — volume normalisation (dataset/brats_dataset/brats.py:26-37, dataset/egd_dataset/egd.py:44-55):
oracle vs the outputs of the reference's own methods (fixture), HIP kernels vs the same fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import input_ref as I

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'input_norm.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


def test_oracle_matches_reference_methods(gold):
    x = torch.from_numpy(gold['x'])
    for v, bz, bm, b01, ez in zip(x, gold['brats_z'], gold['brats_mm'], gold['brats_01'], gold['egd_z']):
        assert np.allclose(I.normalize_data(v, True).numpy(), bz, atol=1e-6)
        assert np.allclose(I.normalize_data(v, False).numpy(), bm, atol=1e-6)
        assert np.allclose(I.min_max_normalize_data(v).numpy(), b01, atol=1e-6)
        assert np.allclose(I.normalize_data(v, True, per_channel=True).numpy(), ez, atol=1e-6)


def test_cpu_input_fails_loudly():
    from vit_ae_plus_plus_amd._abi import VitaeError
    from vit_ae_plus_plus_amd.utils.input_pipeline import normalize_data
    with pytest.raises(VitaeError):
        normalize_data(torch.zeros(1, 1, 4, 4, 4))


@pytest.mark.gpu
def test_hip_normalisation_matches_reference(gold):
    from vit_ae_plus_plus_amd.utils.input_pipeline import min_max_normalize_data, normalize_data
    x = torch.from_numpy(gold['x']).cuda()
    for got, want in ((normalize_data(x, True), gold['brats_z']), (normalize_data(x, False), gold['brats_mm']),
                      (min_max_normalize_data(x), gold['brats_01']), (normalize_data(x, True, per_channel=True), gold['egd_z'])):
        assert np.abs(got.cpu().numpy() - want).max() < 2e-6 * max(1.0, np.abs(want).max())
    # bench-size batch: statistics of the normalised volumes (size-independent property) and the synthetic loader's z-score
    g = torch.Generator(device='cuda').manual_seed(1)
    big = torch.randn(2, 4, 96, 96, 96, device='cuda', generator=g) * 7 - 3
    z = normalize_data(big, True, per_channel=True)
    m = z.double().mean(dim=(2, 3, 4))
    s = z.double().var(dim=(2, 3, 4))
    assert float(m.abs().max()) < 1e-5 and float((s - 1).abs().max()) < 1e-5
    mm = normalize_data(big, False)
    assert float(mm.amin()) == -1.0 and float(mm.amax()) == 1.0
