"""Tiled evaluation on the device (reference test.py:36-71, utils/val_utils.py:50-66).

The reference cuts a test image into `crop_test_imgs_size` tiles at stride = tile size plus one last tile flush with the
border (test.py:47-48), runs the network on the stack of tiles (:57) and averages the overlap (:59-68).  It then accumulates
the INPUT tiles instead of the restored ones (:65, `patched_input_img[cnt]`), so its reported PSNR is that of the degraded
image; this module averages the RESTORED tiles, as the surrounding code intends (`accumulate='input'` reproduces the reference's
literal behaviour).  Everything stays on the GPU: fw_tile_gather cuts the tiles, one batched forward per `max_tiles`, fw_tile_blend
averages the overlap (gather form, no atomics), PSNR as utils/val_utils.py:52-63 and SSIM by fw_ssim7 (skimage's
structural_similarity defaults: 7x7 uniform window, sample covariance, 3-pixel border cropped -- scikit-image itself is not a dependency).
"""
import torch

from .lib import call


def tile_origins(size, tile):
    """test.py:47-48: range(0, size - tile, tile) + [size - tile]."""
    assert size >= tile, 'invalid test image size'
    return list(range(0, size - tile, tile)) + [size - tile]


@torch.no_grad()
def tiled_restore(net, img, tile=128, max_tiles=64, accumulate='restored'):
    """img: f32 [1, C, H, W] on the device -> restored [1, C, H, W] (overlap-averaged)."""
    assert img.dim() == 4 and img.shape[0] == 1 and tile % 8 == 0
    _, C, H, W = img.shape
    ys, xs = tile_origins(H, tile), tile_origins(W, tile)
    dev = img.device
    img = img.contiguous().float()
    yd, xd = torch.tensor(ys, dtype=torch.int32, device=dev), torch.tensor(xs, dtype=torch.int32, device=dev)
    tiles = torch.empty((len(ys) * len(xs), C, tile, tile), dtype=torch.float32, device=dev)
    call('fw_tile_gather', img, yd, xd, tiles, C, H, W, len(ys), len(xs), tile)
    if accumulate == 'input':                                  # test.py:65 as written
        rest = tiles
    else:
        outs = [net(x_query=tiles[i:i + max_tiles], x_key=tiles[i:i + max_tiles]) for i in range(0, tiles.shape[0], max_tiles)]
        rest = torch.cat(outs, 0).float().contiguous()
    out = torch.empty((1, C, H, W), dtype=torch.float32, device=dev)
    call('fw_tile_blend', rest, yd, xd, out, C, H, W, len(ys), len(xs), tile)
    return out


def psnr(restored, clean):
    """utils/val_utils.py:52-63 for a batch: mean over images of 10 log10(1 / mse(clip(a), clip(b)))."""
    a, b = restored.float().clamp(0, 1), clean.float().clamp(0, 1)
    mse = ((a - b) ** 2).flatten(1).mean(1)
    return float((10.0 * torch.log10(1.0 / mse)).mean())


def ssim(restored, clean):
    """utils/val_utils.py:64: mean over images of structural_similarity(clean, restored, data_range=1, channel_axis=2) on the device."""
    a, b = restored.float().contiguous(), clean.float().contiguous()
    assert a.shape == b.shape and a.dim() == 4 and a.is_cuda
    n, C, H, W = a.shape
    out = torch.zeros(n, dtype=torch.float32, device=a.device)
    call('fw_ssim7', a, b, out, n, C, H, W)
    return float((out / (C * (H - 6) * (W - 6))).mean())
