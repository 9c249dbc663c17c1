"""CPU checks of oracle/data_oracle.py (the restatement of test.py:36-71, utils/val_utils.py:50-66 and
utils/dataset_utils.py:122-135 the device kernels of csrc/fw_data.hip are compared with).  SSIM: scikit-image is absent (parity
unpinned); the restatement is anchored by known answers."""
import numpy as np
import torch

import data_oracle as D
import dropout_hash as DH


def test_tiled_restore_grid_and_overlap_average():
    img = torch.arange(3 * 200 * 264, dtype=torch.float32).reshape(1, 3, 200, 264) / 1000.0
    seen = []

    def net(t):
        seen.append(t.shape)
        return t * 2.0 + 1.0                                   # pixel-wise, so the overlap average of tiles equals the map of the image
    out = D.tiled_restore(net, img, 128)
    assert seen == [(6, 3, 128, 128)]                          # rows [0, 72], columns [0, 128, 136] (test.py:47-48)
    assert torch.allclose(out, img * 2.0 + 1.0, atol=1e-5)
    lit = D.tiled_restore(net, img, 128, accumulate='input')   # test.py:65 as written: the degraded input comes back
    assert torch.allclose(lit, img, atol=1e-6)


def test_ssim_known_answers():
    rs = np.random.RandomState(0)
    a = rs.rand(40, 52)
    assert abs(D.ssim_plane(a, a) - 1.0) < 1e-12                                   # identical images
    b = rs.rand(40, 52)
    assert abs(D.ssim_plane(a, b) - D.ssim_plane(b, a)) < 1e-12                    # symmetric
    # constant images u, v: variances 0 -> SSIM = (2uv + C1) / (u^2 + v^2 + C1)
    u, v = 0.3, 0.5
    want = (2 * u * v + 1e-4) / (u * u + v * v + 1e-4)
    assert abs(D.ssim_plane(np.full((20, 20), u), np.full((20, 20), v)) - want) < 1e-12
    # one window exactly: a 7x7 image has a single interior pixel, the statistics are the plain sample moments
    x, y = rs.rand(7, 7), rs.rand(7, 7)
    ux, uy = x.mean(), y.mean()
    vx, vy, vxy = x.var(ddof=1), y.var(ddof=1), ((x - ux) * (y - uy)).sum() / 48.0
    want = ((2 * ux * uy + 1e-4) * (2 * vxy + 9e-4)) / ((ux ** 2 + uy ** 2 + 1e-4) * (vx + vy + 9e-4))
    assert abs(D.ssim_plane(x, y) - want) < 1e-10
    t = torch.from_numpy(np.stack([a, b, a])[None]).float()
    p, s, n = D.psnr_ssim(t, t)
    assert n == 1 and abs(s - 1.0) < 1e-9 and p > 100 or np.isinf(p)


def test_data_augmentation_modes_are_the_eight_symmetries():
    img = np.arange(4 * 4 * 3).reshape(4, 4, 3)
    outs = [D.data_augmentation(img, m) for m in range(8)]
    assert len({o.tobytes() for o in outs}) == 8
    assert np.array_equal(outs[1], img[::-1]) and np.array_equal(outs[4], img[::-1, ::-1])
    assert np.array_equal(outs[2], np.rot90(img)) and np.array_equal(outs[7], np.flipud(np.rot90(img, 3)))


def test_train_batch_semantics():
    rs = np.random.RandomState(1)
    imgs = [rs.randint(0, 256, (3, 150, 170)).astype(np.uint8), rs.randint(0, 256, (3, 128, 200)).astype(np.uint8)]
    deg = [None, rs.randint(0, 256, (3, 128, 200)).astype(np.uint8)]
    rnd = rs.randint(0, 2 ** 31 - 1, (2, 6))
    d1, d2, c1, c2 = D.train_batch(imgs, deg, [25.0, 0.0], rnd, seed=5, site=1000, size=128)
    assert d1.shape == (2, 3, 128, 128) and d1.dtype == np.float32 and 0 <= d1.min() and d1.max() <= 1
    # sample 1 (a degraded image on file): crops are pure gathers of the stored images, same window and mode for both
    assert set(np.unique(np.round(c1[1] * 255)).astype(int)) <= set(np.unique(imgs[1]))
    y0, x0, m = rnd[1][0] % 1, rnd[1][1] % (200 - 127), 1 + rnd[1][2] % 7
    want = D.data_augmentation(np.transpose(deg[1], (1, 2, 0))[y0:y0 + 128, x0:x0 + 128], m)
    assert np.array_equal(np.round(d1[1] * 255).astype(np.uint8), np.transpose(want, (2, 0, 1)))
    # sample 0: noise sigma 25 on the uint8 grid, the SAME noisy image under both crops
    z = D.hashed_normal(5, 1000, 3 * 150 * 170)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs((z ** 3).mean()) < 0.03
    res = (d1[0] - c1[0]) * 255
    assert 15 < res.std() < 25.5                                # clipped N(0, 25)
    assert abs(DH.keep_mask(1, 2, (1000,), 0.5).mean() - 0.5) < 0.06
