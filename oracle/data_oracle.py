"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the callers either side of the model:

  * `tiled_restore`   test.py:36-71  (tile grid, batched forward, overlap average; `accumulate='input'` is the reference's literal
                      line 65, which sums the INPUT tiles; 'restored' is what the surrounding code intends and what the product does)
  * `psnr_ssim`       utils/val_utils.py:50-66.  The reference calls scikit-image (`peak_signal_noise_ratio`, `structural_similarity(
                      ..., data_range=1, channel_axis=2)`), which is NOT installed here and cannot be fetched: PARITY UNPINNED for
                      SSIM.  This restates the published algorithm with skimage's documented defaults (Wang et al. 2004; 7x7 uniform
                      window via scipy.ndimage.uniform_filter -- the very routine skimage calls --, K1 = 0.01, K2 = 0.03, sample
                      covariance NP / (NP - 1), (win - 1) // 2 border pixels cropped, float64) and is anchored by known answers in
                      tests/test_data_oracle.py (identical images -> 1, constant offset closed form, symmetry).
  * `train_batch`     utils/dataset_utils.py:122-135 + utils/image_utils.py:133-182 in numpy on HWC uint8 arrays exactly as the
                      reference writes them (np.flipud / np.rot90, `_crop_patch`, ToTensor); the random draws (crop origins, modes,
                      Gaussian noise) are the product's counter-based integers so that the comparison is deterministic.

Only `tests/`, `smoke()` and `bench.py`'s cpu_baseline may import this module.
"""
import numpy as np
import torch
from scipy.ndimage import uniform_filter

import dropout_hash as DH


# ---------------------------------------------------------------------------------------------------------------- test.py:36-71
def tiled_restore(net_fn, input_img, patch_size=128, accumulate='restored'):
    """input_img: [1, C, H, W] CPU tensor; net_fn(tiles [T, C, p, p]) -> restored tiles.  Follows test.py:41-71 line by line."""
    _, C, H, W = input_img.shape
    assert H >= patch_size and W >= patch_size and patch_size % 8 == 0 and _ == 1
    h_idx_list = list(range(0, H - patch_size, patch_size)) + [H - patch_size]
    w_idx_list = list(range(0, W - patch_size, patch_size)) + [W - patch_size]
    patched = []
    for h_idx in h_idx_list:
        for w_idx in w_idx_list:
            patched.append(input_img[..., h_idx:h_idx + patch_size, w_idx:w_idx + patch_size])
    patched = torch.cat(patched, dim=0)
    restored_tiles = net_fn(patched)
    src = patched if accumulate == 'input' else restored_tiles
    E = torch.zeros(C, H, W).type_as(input_img)
    Wt = torch.zeros_like(E)
    cnt = 0
    for h_idx in h_idx_list:
        for w_idx in w_idx_list:
            E[..., h_idx:h_idx + patch_size, w_idx:w_idx + patch_size].add_(src[cnt])
            Wt[..., h_idx:h_idx + patch_size, w_idx:w_idx + patch_size].add_(torch.ones_like(src[cnt]))
            cnt += 1
    return E.div_(Wt).unsqueeze(0)


# ---------------------------------------------------------------------------------------------------------------- val_utils.py:50-66
def ssim_plane(x, y, data_range=1.0, win=7):
    x, y = x.astype(np.float64), y.astype(np.float64)
    NP = win * win
    cov_norm = NP / (NP - 1.0)
    ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
    uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return S[pad:-pad, pad:-pad].mean()


def psnr_ssim(recovered, clean):
    """-> (mean PSNR, mean SSIM, N) over the batch, inputs [N, C, H, W] clipped to [0, 1] (val_utils.py:50-66)."""
    a = np.clip(recovered.detach().cpu().numpy(), 0, 1).astype(np.float64)
    b = np.clip(clean.detach().cpu().numpy(), 0, 1).astype(np.float64)
    ps, ss = 0.0, 0.0
    for i in range(a.shape[0]):
        mse = np.mean((a[i] - b[i]) ** 2)
        ps += 10.0 * np.log10(1.0 / mse)
        ss += np.mean([ssim_plane(b[i, c], a[i, c]) for c in range(a.shape[1])])      # channel_axis: mean of the per-channel values
    return ps / a.shape[0], ss / a.shape[0], a.shape[0]


# ---------------------------------------------------------------------------------------------------------------- dataset_utils.py:122-135
def data_augmentation(image, mode):
    """utils/image_utils.py:133-160 on an HWC numpy array."""
    if mode == 0:
        return image
    out = np.rot90(image, k=mode // 2) if mode // 2 else image
    return np.flipud(out) if mode % 2 == 1 else out


def hashed_normal(seed, site, n):
    """The product's counter-based N(0, 1) for flat pixel indices 0..n-1 (csrc/fw_data.hip: Box-Muller of two hashed u32)."""
    key = np.uint64(DH.site_key(seed, site))
    idx = np.arange(n, dtype=np.uint64)
    r1 = DH._hash32((idx & DH.M32) ^ key)
    r2 = DH._hash32((r1 + np.uint64(0x9E3779B9)) & DH.M32)
    u1 = (r1.astype(np.float64) + 1.0) * 2.3283064365386963e-10
    u2 = r2.astype(np.float64) * 2.3283064365386963e-10
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def train_batch(images_u8, degraded_u8, sigma, rnd, seed, site, size):
    """images_u8: list of uint8 [3, H, W] arrays (channels first, as they lie in HBM); degraded_u8[i]: array or None; sigma[i]: noise
    level when there is no degraded image; rnd: int [B][6] = (y1, x1, m1, y2, x2, m2) raw draws; -> four float32 [B, 3, S, S] arrays
    (degrad_patch_1, degrad_patch_2, clean_patch_1, clean_patch_2), restating dataset_utils.py:122-135 on HWC arrays."""
    outs = [[], [], [], []]
    for b, gt_chw in enumerate(images_u8):
        gt_img = np.transpose(gt_chw, (1, 2, 0))                                    # HWC like np.array(Image.open(...))
        H, W = gt_img.shape[:2]
        if degraded_u8[b] is not None:
            input_img = np.transpose(degraded_u8[b], (1, 2, 0))
        elif sigma[b] > 0:
            z = hashed_normal(seed, site + b, 3 * H * W).reshape(3, H, W).transpose(1, 2, 0)   # the kernel indexes the noise by CHW pixel
            input_img = np.clip(gt_img + z * sigma[b], 0, 255).astype(np.uint8)      # dataset_utils.py:126
        else:
            input_img = gt_img
        for v in range(2):
            ry, rx, rm = (int(t) for t in rnd[b][3 * v:3 * v + 3])
            y0, x0, mode = ry % (H - size + 1), rx % (W - size + 1), 1 + rm % 7
            d = data_augmentation(input_img[y0:y0 + size, x0:x0 + size], mode).copy()
            c = data_augmentation(gt_img[y0:y0 + size, x0:x0 + size], mode).copy()
            outs[v].append(np.transpose(d, (2, 0, 1)).astype(np.float32) / 255.0)    # ToTensor
            outs[2 + v].append(np.transpose(c, (2, 0, 1)).astype(np.float32) / 255.0)
    return tuple(np.stack(o, 0) for o in outs)
