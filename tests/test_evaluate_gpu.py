"""Tiled evaluation on the device (fwair/evaluate.py) against tile-by-tile forwards, and its PSNR against the oracle's."""
import pytest
import torch

import airnet_oracle as O
from helpers import make_opt

pytestmark = pytest.mark.gpu


def test_tiled_restore_matches_tile_by_tile():
    from fwair import evaluate as EV
    from net.model import AirNet
    torch.manual_seed(3)
    net = AirNet(make_opt('all3')).to('cuda').eval()
    img = torch.rand(1, 3, 200, 264, device='cuda')
    out = EV.tiled_restore(net, img, tile=128, max_tiles=3)
    assert out.shape == img.shape and torch.isfinite(out).all()
    ys, xs = EV.tile_origins(200, 128), EV.tile_origins(264, 128)
    assert ys == [0, 72] and xs == [0, 128, 136]                       # test.py:47-48
    acc, wgt = torch.zeros(3, 200, 264, device='cuda'), torch.zeros(1, 200, 264, device='cuda')
    with torch.no_grad():
        for y in ys:
            for x in xs:
                t = img[:, :, y:y + 128, x:x + 128].contiguous()
                acc[:, y:y + 128, x:x + 128] += net(x_query=t, x_key=t)[0].float()
                wgt[:, y:y + 128, x:x + 128] += 1
    ref = (acc / wgt).unsqueeze(0)
    assert float((out - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    clean = torch.rand(1, 3, 200, 264)
    assert abs(EV.psnr(out.cpu(), clean) - float(O.psnr(out.cpu(), clean))) < 1e-3
