"""Synthetic training batches: low-frequency cosines + rectangles quantised to uint8, denoising
degradation as utils/dataset_utils.py:126 (clip(clean*255 + sigma*randn, 0, 255) as uint8 / 255, two independent draws)."""
import numpy as np
import torch


def synth_batch(B, size, sigma, seed, device):
    """-> (clean, degraded_1, degraded_2), f32 [B, 3, size, size] on `device`."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    clean = np.zeros((B, 3, size, size), np.float32)
    for b in range(B):
        for c in range(3):
            img = np.zeros((size, size), np.float32)
            for _ in range(8):
                fx, fy, ph, a = rs.uniform(0, 4), rs.uniform(0, 4), rs.uniform(0, 6.28), rs.uniform(0.2, 1)
                img += a * np.cos(6.2832 * (fx * xx + fy * yy) + ph)
            for _ in range(4):
                x0, y0 = rs.randint(0, size - 8, 2)
                w, h = rs.randint(8, size // 2, 2)
                img[y0:y0 + h, x0:x0 + w] += rs.uniform(-1, 1)
            img = (img - img.min()) / max(img.max() - img.min(), 1e-6)
            clean[b, c] = np.round(img * 255) / 255

    def noisy():
        return np.clip(clean * 255 + sigma * rs.randn(*clean.shape), 0, 255).astype(np.uint8).astype(np.float32) / 255
    t = lambda a: torch.from_numpy(a).to(device)
    return t(clean), t(noisy()), t(noisy())


def synth_task_batch(B, size, tasks, seed, device):
    """Mixed-degradation batch (BASELINE configs[2] / [3]): sample i carries task tasks[i % len(tasks)], exactly as the
    reference's dataset cycles `de_type` item by item (dataset_utils.py:99).  The degraded image is made once per sample on the
    device; the two views are two independent flip / rotation modes of it (the dataset's two augmented crops, :131-132) and
    the clean target follows the first view.  -> (clean_1, degraded_1, degraded_2)."""
    from . import augment as A
    clean, _, _ = synth_batch(B, size, 0, seed, device)
    g = torch.Generator(device='cpu'); g.manual_seed(seed)
    cu8 = (clean * 255.0).round().to(torch.uint8)
    c1, d1, d2 = [], [], []
    for i in range(B):
        deg = A.degrade(cu8[i], tasks[i % len(tasks)], g)
        m1, m2 = (int(torch.randint(1, 8, (1,), generator=g)) for _ in range(2))
        c1.append(A.augment(cu8[i], m1)); d1.append(A.augment(deg, m1)); d2.append(A.augment(deg, m2))
    f = lambda ts: torch.stack(ts, 0).float().div_(255.0).contiguous()
    return f(c1), f(d1), f(d2)
