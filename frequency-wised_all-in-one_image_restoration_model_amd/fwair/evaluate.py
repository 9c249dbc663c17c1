"""Tiled evaluation on the device (reference test.py:36-71).

The reference cuts a test image into `crop_test_imgs_size` tiles at stride = tile size plus one last tile flush with the
border (test.py:47-48), runs the network on the stack of tiles (:57) and averages the overlap (:59-68).  It then accumulates
the INPUT tiles instead of the restored ones (:65, `patched_input_img[cnt]`), so its reported PSNR is that of the degraded
image; this module averages the RESTORED tiles, as the surrounding code intends.  Everything stays on the GPU: one batched
forward for all tiles of an image (chunked by `max_tiles`), overlap weights by index arithmetic, PSNR as utils/val_utils.py:52-63
(per image, clip to [0, 1], 10 log10(1 / mse)).  SSIM needs scikit-image, which is not a dependency of this package.
"""
import torch


def tile_origins(size, tile):
    """test.py:47-48: range(0, size - tile, tile) + [size - tile]."""
    assert size >= tile, 'invalid test image size'
    return list(range(0, size - tile, tile)) + [size - tile]


@torch.no_grad()
def tiled_restore(net, img, tile=128, max_tiles=64):
    """img: f32 [1, C, H, W] on the device -> restored [1, C, H, W] (overlap-averaged)."""
    assert img.dim() == 4 and img.shape[0] == 1 and tile % 8 == 0
    _, C, H, W = img.shape
    ys, xs = tile_origins(H, tile), tile_origins(W, tile)
    tiles = torch.stack([img[0, :, y:y + tile, x:x + tile] for y in ys for x in xs], 0)
    outs = [net(x_query=tiles[i:i + max_tiles], x_key=tiles[i:i + max_tiles]) for i in range(0, tiles.shape[0], max_tiles)]
    rest = torch.cat(outs, 0).float()
    acc = torch.zeros((C, H, W), dtype=torch.float32, device=img.device)
    wgt = torch.zeros((1, H, W), dtype=torch.float32, device=img.device)
    k = 0
    for y in ys:
        for x in xs:
            acc[:, y:y + tile, x:x + tile] += rest[k]
            wgt[:, y:y + tile, x:x + tile] += 1.0
            k += 1
    return (acc / wgt).unsqueeze(0)


def psnr(restored, clean):
    """utils/val_utils.py:52-63 for a batch: mean over images of 10 log10(1 / mse(clip(a), clip(b)))."""
    a, b = restored.float().clamp(0, 1), clean.float().clamp(0, 1)
    mse = ((a - b) ** 2).flatten(1).mean(1)
    return float((10.0 * torch.log10(1.0 / mse)).mean())
