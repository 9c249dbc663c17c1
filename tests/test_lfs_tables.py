"""Host logic of the LFS band filter: the partial-DFT panel chain (numpy emulation of csrc/fw_attn.hip
band_filter) must reproduce band 1 of the oracle's FFT-based decomposition, and the affine identity
P' = (1+l2) P - l2/64 + (l1-l2) B1(P) must equal the reference's  P + l1*band1 + l2*band2."""
import numpy as np
import torch

import airnet_oracle as O
from fwair import lfs


def test_panel_chain_matches_fft_band1():
    torch.manual_seed(0)
    P = torch.softmax(torch.randn(3, 2, 64, 64, dtype=torch.float64) * 2, -1)
    bands = O.frequency_decompose(P, 'frequency_decompose_1', 0.5, 64, 64, True)
    masks = lfs.band_masks_shifted('frequency_decompose_1', 0.5, 64, 64)
    for a, b in zip(masks, O.band_masks('frequency_decompose_1', 0.5, 64, 64)):
        assert torch.equal(a, b)
    panels, Mw = lfs.build_panels(masks[1].numpy())
    out = lfs.emulate_filter(P.numpy(), panels, Mw.astype(np.float64))
    assert np.abs(out - bands[1].numpy()).max() < 1e-12
    # band 0 is the mean = 1/64 for a softmax map; the bands partition the plane
    assert np.abs(bands[0].numpy() - 1 / 64).max() < 1e-12
    assert np.abs(bands.sum(0).numpy() - P.numpy()).max() < 1e-12
    l1, l2 = 0.37, -0.21
    ref = P + l1 * bands[1] + l2 * bands[2]
    mine = (1 + l2) * P.numpy() - l2 / 64 + (l1 - l2) * out
    assert np.abs(mine - ref.numpy()).max() < 1e-12


def test_filter_is_self_adjoint():
    rng = np.random.RandomState(1)
    masks = lfs.band_masks_shifted('frequency_decompose_1', 0.5, 64, 64)
    panels, Mw = lfs.build_panels(masks[1].numpy())
    A, B = rng.randn(64, 64), rng.randn(64, 64)
    fa = lfs.emulate_filter(A, panels, Mw.astype(np.float64))
    fb = lfs.emulate_filter(B, panels, Mw.astype(np.float64))
    assert abs((fa * B).sum() - (A * fb).sum()) < 1e-10


def test_table_size_matches_library():
    from fwair import lib
    masks = lfs.band_masks_shifted('frequency_decompose_1', 0.5, 64, 64)
    panels, _ = lfs.build_panels(masks[1].numpy())
    assert panels.size == lib.lib().fw_attn_lfs_table_elems()
