"""Host-side constants for the learned-frequency-selection (LFS) band filter of fw_attn_fwd/bwd.

The decoder's attention maps are 64x64; the reference decomposes them with
FrequencyDecompose('frequency_decompose_1', 1/(nb-1), 64, 64) (net/decoder_Uformer.py:170) and adds
lambda_i * band_i (:275-288).  For nb = 3 only band 1 -- the disc 0 < |f| <= r/2 -- needs a transform
(band 0 is the mean, band 2 the remainder).  csrc/fw_attn.hip evaluates it as a partial DFT restricted
to |fu| <= 22, 0 <= fv <= 22; this module builds its cos/sin panels and the weighted mask.
Panel order and shapes must match the OFF_* constants of csrc/fw_attn.hip.
"""
import math

import numpy as np
import torch

N = 64
FU = 22            # |fu| <= FU  -> 45 rows, padded to NU
NU, NV = 48, 32


def band_masks_shifted(kind, size, h, w):
    """Boolean band masks in fftshift-ed coordinates, arithmetic as net/utils/frequency_decompose.py:17-26,37-49,79-88
    (int64 grid -> float32 sqrt, so the <= / < edge bins fall on the reference's side)."""
    Y = torch.arange(h).unsqueeze(1)
    X = torch.arange(w).unsqueeze(0)
    num_bands = math.floor(1. / size + 0.1)
    center = torch.tensor([int(w / 2), int(h / 2)])
    dist = torch.sqrt((X - center[0]) ** 2 + (Y - center[1]) ** 2)
    max_radius = torch.sqrt(center[0] ** 2 + center[1] ** 2)
    last = torch.zeros((h, w), dtype=torch.bool)
    out = []
    if kind == 'frequency_decompose':
        for sz in torch.linspace(size, 1, num_bands):
            mask = (dist <= max_radius * sz) if sz == 1.0 else (dist < max_radius * sz)
            out.append(mask ^ last)
            last = mask
    elif kind == 'frequency_decompose_1':
        for sz in torch.linspace(0, 1, num_bands + 1):
            mask = dist <= max_radius * sz
            out.append(mask ^ last)
            last = mask
    else:
        raise ValueError(kind)
    return out


def build_panels(mask_shifted):
    """mask_shifted: bool [64][64] (rows = row frequency + 32, cols = column frequency + 32).
    Returns (panels float64 1-D in OFF_* order, Mw float32 [48][32])."""
    m = np.asarray(mask_shifted, dtype=bool)
    assert m.shape == (N, N)
    fy, fx = np.nonzero(m)
    fy = fy - N // 2
    fx = fx - N // 2
    if np.abs(fy).max() > FU or np.abs(fx).max() > FU:
        raise NotImplementedError('band mask exceeds the |f| <= 22 support the HIP filter is built for')
    # Hermitian symmetry (real filter) -- required by the half-spectrum evaluation
    mm = m[1:, 1:]
    assert np.array_equal(mm, mm[::-1, ::-1]) and not m[0].any() and not m[:, 0].any()
    j = np.arange(N)
    v = np.arange(NV)
    u = np.arange(NU)
    fu = u - FU
    vv = (v < FU + 1).astype(np.float64)
    uu = (u < 2 * FU + 1).astype(np.float64)
    ang_vj = 2 * np.pi * np.outer(v, j) / N            # [v][j]
    ang_ui = 2 * np.pi * np.outer(fu, j) / N           # [u][i]
    C2 = np.cos(ang_vj) * vv[:, None]
    S2N = -np.sin(ang_vj) * vv[:, None]
    CU = np.cos(ang_ui) * uu[:, None]
    SU = np.sin(ang_ui) * uu[:, None]
    CH = np.zeros((N, N)); SH = np.zeros((N, N))
    CH[:, :NU] = CU.T
    SH[:, :NU] = SU.T
    GC = C2.T.copy()                                    # [j][v]
    GSN = S2N.T.copy()
    panels = np.concatenate([a.reshape(-1) for a in (C2, S2N, CU, SU, -SU, CH, SH, -SH, GC, GSN)])
    Mw = np.zeros((NU, NV), dtype=np.float32)
    for a in range(2 * FU + 1):
        for b in range(FU + 1):
            if m[a - FU + N // 2, b + N // 2]:
                Mw[a, b] = (1.0 if b == 0 else 2.0) / (N * N)
    return panels, Mw


def emulate_filter(P, panels, Mw):
    """numpy restatement of band_filter() in csrc/fw_attn.hip (same products, same order); P: [..., 64, 64]."""
    o = 0

    def take(r, c):
        nonlocal o
        a = panels[o:o + r * c].reshape(r, c)
        o += r * c
        return a
    C2, S2N, CU, SU, SUN, CH, SH, SHN, GC, GSN = (take(32, 64), take(32, 64), take(48, 64), take(48, 64), take(48, 64),
                                                    take(64, 64), take(64, 64), take(64, 64), take(64, 32), take(64, 32))
    Tr = P @ C2.T                                       # [i][v]
    Ti = P @ S2N.T
    Xr = CU @ Tr + SU @ Ti                              # [u][v]
    Xi = CU @ Ti + SUN @ Tr
    Yr = Xr * Mw
    Yi = Xi * Mw
    Ypr = np.zeros(Yr.shape[:-2] + (64, 32)); Ypi = np.zeros_like(Ypr)
    Ypr[..., :48, :] = Yr
    Ypi[..., :48, :] = Yi
    ZrT = np.swapaxes(Ypr, -1, -2) @ CH.T + np.swapaxes(Ypi, -1, -2) @ SHN.T     # [v][i]
    ZiT = np.swapaxes(Ypi, -1, -2) @ CH.T + np.swapaxes(Ypr, -1, -2) @ SH.T
    outT = GC @ ZrT + GSN @ ZiT                         # [j][i]
    return np.swapaxes(outT, -1, -2)


_cache = {}


def device_table(dtype, device, nb=3):
    """Byte buffer for fw_attn_*: panels in `dtype` followed by the f32 mask.  nb = number of bands (3)."""
    key = (dtype, str(device), nb)
    if key not in _cache:
        masks = band_masks_shifted('frequency_decompose_1', 1. / (nb - 1), N, N)
        assert nb == 3, 'only the 3-band decomposition needs (and has) a HIP band filter'
        panels, Mw = build_panels(masks[1].numpy())
        p = torch.from_numpy(panels).to(dtype).contiguous()
        raw = torch.cat([p.view(torch.uint8).reshape(-1), torch.from_numpy(Mw).reshape(-1).view(torch.uint8)])
        _cache[key] = raw.to(device)
    return _cache[key]
