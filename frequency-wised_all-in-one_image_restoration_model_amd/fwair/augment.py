"""Input pipeline on the device: the per-sample work of utils/dataset_utils.py:122-135 -- Gaussian noise
synthesis on the uint8 grid, two independent random crops of the same degraded image, one of the 7 flip / rotation modes of
utils/image_utils.py:133-160 per crop -- as batched tensor ops, so that it runs on the GPU next to the model instead of in
DataLoader workers.  Images are channels-first ([..., C, H, W]); the reference's numpy code is HWC (flipud = flip H,
rot90 = counter-clockwise in the H-W plane), which maps to dims (-2, -1) here."""
import math

import torch

from .lib import call


def augment(x, mode):
    """utils/image_utils.py:133-160 `data_augmentation(image, mode)` for channels-first tensors."""
    if mode == 0:
        return x
    k, flip = mode // 2, mode % 2 == 1          # 1: flipud | 2: rot90 | 3: rot90 + flipud | 4: rot180 | 5: +flipud | 6: rot270 | 7: +flipud
    out = torch.rot90(x, k, dims=(-2, -1)) if k else x
    return out.flip(-2) if flip else out


def add_noise(clean_u8, sigma, generator=None):
    """dataset_utils.py:126: clip(gt + randn * sigma, 0, 255).astype(uint8).  clean_u8: uint8 or float tensor on the 0..255 grid."""
    g = clean_u8.float()
    if generator is not None and generator.device != g.device:      # a host generator driving device data: draw there, move
        n = torch.randn(g.shape, device=generator.device, generator=generator).to(g.device)
    else:
        n = torch.randn(g.shape, device=g.device, generator=generator)
    return (g + n * float(sigma)).clamp_(0, 255).to(torch.uint8)          # float -> uint8 truncates, like numpy's astype


def add_rain(clean_u8, generator=None, streaks=None, length=None, angle_deg=None, strength=0.7):
    """Synthetic stand-in for the rain pairs the reference only loads from disk (dataset_utils.py:93-95,129):
    additive bright line streaks at 45 +- 15 degrees, drawn as `length` shifted copies of a sparse seed map.  uint8 in / out."""
    g = clean_u8.float()
    H, W = g.shape[-2:]
    dev = g.device
    rnd = lambda *shape: torch.rand(shape, generator=generator, device='cpu')
    ang = float(angle_deg) if angle_deg is not None else 30.0 + 30.0 * float(rnd(1))
    n = int(streaks) if streaks is not None else max(1, H * W // 400)
    L = int(length) if length is not None else max(4, H // 8)
    ys = (rnd(n) * H).long().clamp_(max=H - 1)
    xs = (rnd(n) * W).long().clamp_(max=W - 1)
    seed = torch.zeros((H, W), dtype=torch.float32)
    seed[ys, xs] = 0.5 + 0.5 * rnd(n)
    seed = seed.to(dev)
    dy, dx = math.sin(math.radians(ang)), math.cos(math.radians(ang))
    layer = torch.zeros((H, W), dtype=torch.float32, device=dev)
    for t in range(L):                                     # a streak = the seed pixel smeared along (dy, dx), fading towards its tail
        layer = torch.maximum(layer, torch.roll(seed, shifts=(int(round(t * dy)), int(round(t * dx))), dims=(0, 1)) * (1.0 - t / L))
    return (g + 255.0 * strength * layer).clamp_(0, 255).to(torch.uint8)


def add_haze(clean_u8, generator=None, beta=None, airlight=0.8):
    """Synthetic stand-in for the haze pairs: I = J t + A (1 - t), t = exp(-beta * depth), depth = a vertical ramp
    (far at the top of the frame), A = 0.8.  uint8 in / out."""
    g = clean_u8.float() / 255.0
    H = g.shape[-2]
    b = float(beta) if beta is not None else 0.6 + 1.2 * float(torch.rand((1,), generator=generator, device='cpu'))
    depth = torch.linspace(1.0, 0.1, H, device=g.device).view(H, 1)
    t = torch.exp(-b * depth)
    return ((g * t + airlight * (1.0 - t)) * 255.0).clamp_(0, 255).to(torch.uint8)


def degrade(clean_u8, task, generator=None):
    """The degraded image of one `--de_type` entry.  denoising_<sigma> as dataset_utils.py:123-126 (sigma 0 = a random choice of
    15 / 25 / 50); deraining / dehazing: the synthetic stand-ins above (the reference reads those pairs from disk)."""
    if task.startswith('denoising'):
        sigma = int(task.split('_')[-1])
        if sigma == 0:
            sigma = (15, 25, 50)[int(torch.randint(0, 3, (1,), generator=generator, device='cpu'))]
        return add_noise(clean_u8, sigma, generator)
    if task == 'deraining':
        return add_rain(clean_u8, generator)
    if task == 'dehazing':
        return add_haze(clean_u8, generator)
    raise ValueError(f'unknown de_type {task!r}')


def crop_pair(degraded, clean, size, generator=None):
    """dataset_utils.py `_crop_patch`: one random window, the same for the degraded and the clean image ([C, H, W] each)."""
    H, W = clean.shape[-2:]
    y = int(torch.randint(0, H - size + 1, (1,), generator=generator, device='cpu'))
    x = int(torch.randint(0, W - size + 1, (1,), generator=generator, device='cpu'))
    return degraded[..., y:y + size, x:x + size], clean[..., y:y + size, x:x + size]


def training_pair(clean_u8, size, sigma=None, degraded_u8=None, generator=None):
    """One dataset item (dataset_utils.py:122-135) from a clean uint8 image [3, H, W] on the device:
    -> (degrad_patch_1, degrad_patch_2, clean_patch_1, clean_patch_2), f32 in [0, 1] as ToTensor would give.
    Denoising: the degraded image is synthesised once (sigma), then cropped twice; other tasks pass `degraded_u8`."""
    if degraded_u8 is None:
        degraded_u8 = add_noise(clean_u8, sigma, generator)
    out = []
    for _ in range(2):
        d, c = crop_pair(degraded_u8, clean_u8, size, generator)
        mode = int(torch.randint(1, 8, (1,), generator=generator, device='cpu'))          # random_augmentation: 1..7, never 0
        out.append((augment(d, mode).float() / 255.0, augment(c, mode).float() / 255.0))
    (d1, c1), (d2, c2) = out
    return d1.contiguous(), d2.contiguous(), c1.contiguous(), c2.contiguous()


def training_batch(images_u8, size, sigmas, generator=None):
    """A batch: images_u8 = list of [3, H, W] uint8 device tensors (sizes may differ); sigmas = per image a noise level, or a
    `--de_type` name ('denoising_25', 'deraining', 'dehazing': the dataset cycles its tasks item by item, dataset_utils.py:99)."""
    items = []
    for img, s in zip(images_u8, sigmas):
        if isinstance(s, str):
            items.append(training_pair(img, size, degraded_u8=degrade(img, s, generator), generator=generator))
        else:
            items.append(training_pair(img, size, sigma=s, generator=generator))
    return tuple(torch.stack(t, 0) for t in zip(*items))


# ---------------------------------------------------------------------------------------------------------------
# the whole batch in ONE launch (csrc/fw_data.hip: fw_train_batch) -- no host round trip per crop
# ---------------------------------------------------------------------------------------------------------------
_TASK_SITE = 0x7A000000


class DeviceBatcher:
    """Training batches from uint8 images resident in HBM at device rate.

    images_u8: list of [3, H, W] uint8 device tensors (sizes may differ).  tasks[i]: a noise level, or a `--de_type` name; denoising is
    synthesised INSIDE the kernel (counter-based N(0, 1) per pixel of the full image, so the two crops of a sample share their noise
    exactly as the reference's two crops of one noisy image do, dataset_utils.py:126,131-132); 'deraining' / 'dehazing' samples carry
    a degraded image made once at construction (the reference reads those pairs from disk, :93-95,129).
    `batch(indices)`: crop origins and flip / rotation modes of the whole batch from ONE device RNG call, one kernel launch, no sync.
    The pointer table of a given index list is built once and cached (a training loop cycles a fixed schedule of index lists)."""

    def __init__(self, images_u8, tasks, size, generator=None):
        from . import functional as Fn
        self.size = int(size)
        self.images = [im.contiguous() for im in images_u8]
        self.dev = self.images[0].device
        self.degraded, self.sigma = [], []
        for im, t in zip(self.images, tasks):
            assert im.dtype == torch.uint8 and im.dim() == 3 and im.shape[0] == 3 and min(im.shape[1:]) >= self.size
            if isinstance(t, str) and not t.startswith('denoising'):
                self.degraded.append(degrade(im, t, generator).contiguous()); self.sigma.append(0.0)
            else:
                sg = float(t.split('_')[-1]) if isinstance(t, str) else float(t)
                self.degraded.append(None); self.sigma.append(sg)
        self.random_sigma = [s == 0.0 and d is None and isinstance(t, str) for s, d, t in zip(self.sigma, self.degraded, tasks)]
        self.seed = Fn.dropout_seed(self.dev)
        self._tables = {}
        self._calls = 0

    def _table(self, idx):
        key = tuple(idx)
        t = self._tables.get(key)
        if t is None:
            rows = [[self.images[i].data_ptr(), self.degraded[i].data_ptr() if self.degraded[i] is not None else 0,
                     self.images[i].shape[1], self.images[i].shape[2]] for i in idx]
            sig = torch.tensor([self.sigma[i] for i in idx], dtype=torch.float32)
            rs = torch.tensor([self.random_sigma[i] for i in idx], dtype=torch.bool)
            t = self._tables[key] = (torch.tensor(rows, dtype=torch.int64).to(self.dev), sig.to(self.dev), rs.to(self.dev))
        return t

    def batch(self, indices, generator=None):
        """-> (degrad_patch_1, degrad_patch_2, clean_patch_1, clean_patch_2), f32 [B, 3, S, S] in [0, 1]."""
        B, S = len(indices), self.size
        tab, sigma, rs = self._table(indices)
        rnd = torch.randint(0, 2 ** 31 - 1, (B, 7), dtype=torch.int32, device=self.dev, generator=generator)
        if bool(any(self.random_sigma[i] for i in indices)):          # denoising_0: sigma drawn from {15, 25, 50} per sample (:124-125)
            choice = torch.tensor([15.0, 25.0, 50.0], device=self.dev)[(rnd[:, 6] % 3).long()]
            sigma = torch.where(rs, choice, sigma)
        out = [torch.empty((B, 3, S, S), dtype=torch.float32, device=self.dev) for _ in range(4)]
        self._calls += 1
        site = _TASK_SITE + (self._calls & 0xFFFF) * 256
        call('fw_train_batch', tab, rnd[:, :6].contiguous(), sigma.contiguous(), self.seed, site, *out, B, S)
        self.last_rnd, self.last_site, self.last_sigma = rnd, site, sigma
        return tuple(out)
