"""Input pipeline on the device: the per-sample work of utils/dataset_utils.py:122-135 -- Gaussian noise
synthesis on the uint8 grid, two independent random crops of the same degraded image, one of the 7 flip / rotation modes of
utils/image_utils.py:133-160 per crop -- as batched tensor ops, so that it runs on the GPU next to the model instead of in
DataLoader workers.  Images are channels-first ([..., C, H, W]); the reference's numpy code is HWC (flipud = flip H,
rot90 = counter-clockwise in the H-W plane), which maps to dims (-2, -1) here."""
import torch


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
    n = torch.randn(g.shape, device=g.device, generator=generator)
    return (g + n * float(sigma)).clamp_(0, 255).to(torch.uint8)          # float -> uint8 truncates, like numpy's astype


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
    """A batch: images_u8 = list of [3, H, W] uint8 device tensors (sizes may differ), sigmas = one noise level per image."""
    items = [training_pair(img, size, sigma=s, generator=generator) for img, s in zip(images_u8, sigmas)]
    return tuple(torch.stack(t, 0) for t in zip(*items))
