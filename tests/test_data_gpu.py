"""Device kernels of the callers either side of the model (csrc/fw_data.hip) against the CPU oracle
(oracle/data_oracle.py: test.py:36-71, utils/val_utils.py:50-66, utils/dataset_utils.py:122-135 restated in numpy / torch):
  * fw_train_batch: a mixed-task batch from uint8 images of different sizes -- crops, flip / rotation modes, ToTensor scaling
    bit-exact; in-kernel Gaussian noise on the uint8 grid equal to the oracle's f64 evaluation of the same counter-based draws up to
    one grey level on a < 1e-3 fraction of the pixels (f32 vs f64 Box-Muller at a truncation boundary);
  * tiled evaluation (fw_tile_gather / fw_tile_blend around the HIP model) against the oracle's test.py loops around the ORACLE
    network -- not against the same network tile by tile;
  * fw_ssim7 against the oracle's skimage-default SSIM (parity unpinned against skimage itself: not installed);
  * the input pipeline's rate: no host synchronisation inside, far above the 1 800 images/s/GPU the step would need at its roofline."""
import time

import numpy as np
import pytest
import torch

import airnet_oracle as O
import data_oracle as D
from helpers import close, make_opt, schema, synth_batch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _images(seed, sizes):
    rs = np.random.RandomState(seed)
    out = []
    for (h, w) in sizes:
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        img = np.stack([127 + 90 * np.sin(xx / (7 + c) + yy / (11 + 2 * c)) + rs.randn(h, w) * 12 for c in range(3)])
        out.append(np.clip(img, 0, 255).astype(np.uint8))
    return out


def test_train_batch_vs_oracle():
    from fwair import augment as A
    from fwair import functional as Fn
    imgs = _images(3, [(150, 170), (128, 200), (256, 256), (129, 128), (321, 200)])
    tasks = [25, 'denoising_15', 'deraining', 'dehazing', 50]
    dev_imgs = [torch.from_numpy(i).to(DEV) for i in imgs]
    Fn.set_dropout_seed(31337, DEV, frozen=True)
    try:
        bt = A.DeviceBatcher(dev_imgs, tasks, 128, generator=torch.Generator().manual_seed(5))
        idx = [4, 0, 1, 2, 3, 0, 2]
        g = torch.Generator(device=DEV).manual_seed(9)
        out = [t.cpu().numpy() for t in bt.batch(idx, generator=g)]
        rnd = bt.last_rnd.cpu().numpy()
        deg = [bt.degraded[i].cpu().numpy() if bt.degraded[i] is not None else None for i in idx]
        ref = D.train_batch([imgs[i] for i in idx], deg, [bt.sigma[i] for i in idx], rnd[:, :6], 31337, bt.last_site, 128)
    finally:
        Fn.set_dropout_seed(1, DEV, frozen=False)
    for name, a, b in zip(('degrad_patch_1', 'degrad_patch_2', 'clean_patch_1', 'clean_patch_2'), out, ref):
        assert a.shape == b.shape == (7, 3, 128, 128)
        if name.startswith('clean'):
            assert np.array_equal(a, b), name                                       # pure gathers: bit-exact
            continue
        diff = np.abs(a - b) * 255
        assert diff.max() <= 1.0 + 1e-4, f'{name}: off by {diff.max():.3f} grey levels'
        assert (diff > 1e-4).mean() < 1e-3, f'{name}: {(diff > 1e-4).mean():.2e} of the pixels differ'
        for s in range(7):
            if deg[s] is not None:
                assert np.array_equal(a[s], b[s]), f'{name}[{s}]: degraded image on file -> bit-exact gather'
    # same sample twice in the batch (index 0 at positions 1 and 5): different crops / modes, and different noise draws per batch slot
    assert not np.array_equal(out[0][1], out[0][5])


def test_input_pipeline_rate_and_no_host_sync():
    from fwair import augment as A
    imgs = [torch.from_numpy(i).to(DEV) for i in _images(4, [(256, 256)] * 64)]
    bt = A.DeviceBatcher(imgs, [25] * 64, 128)
    idx = [list(range(k, k + 16)) for k in range(0, 64, 16)]
    g = torch.Generator(device=DEV).manual_seed(1)
    for i in range(8):
        bt.batch(idx[i % 4], generator=g)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    with torch.autograd.profiler.profile(enabled=False):
        for i in range(n):
            bt.batch(idx[i % 4], generator=g)
    t_issue = time.perf_counter() - t0                                 # host time to ISSUE n batches: no sync inside means this is launch cost only
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    rate = 16 * n / t_all
    print(f'device input pipeline: {rate:.0f} images/s ({t_all / n * 1e6:.0f} us per batch of 16; host issue {t_issue / n * 1e6:.0f} us)')
    assert rate > 18000, 'the input pipeline must not bound a 1 800 images/s training step'


def _seeded_net():
    from net.model import AirNet
    opt = make_opt('all3')
    net = AirNet(opt)
    st = O.fill_state_seeded(schema('all3'))
    sd = net.state_dict()
    for k in sd:
        if st.get(k) is not None and sd[k].is_floating_point():
            sd[k] = st[k]
    net.load_state_dict(sd)
    return net.to(DEV).eval(), opt, st


def test_tiled_restore_vs_oracle_network():
    """test.py:36-71 around the model: the product (tile kernels + HIP network) against the oracle's loops around the ORACLE network."""
    from fwair import evaluate as EV
    net, opt, st = _seeded_net()
    clean, q, _ = synth_batch(1, 256, 'tiles.')
    img = q[:, :, :200, :248].contiguous()                            # rows [0, 72], columns [0, 120]: both axes overlap
    assert EV.tile_origins(200, 128) == [0, 72] and EV.tile_origins(248, 128) == [0, 120]
    out = EV.tiled_restore(net, img.to(DEV), tile=128, max_tiles=3)
    with torch.no_grad():
        ref = D.tiled_restore(lambda t: O.airnet_forward(st, opt, t, t, False), img, 128)
    close(out, ref, 1e-4, 'tiled restore vs oracle (restored tiles)')
    lit = EV.tiled_restore(net, img.to(DEV), tile=128, accumulate='input')
    close(lit, D.tiled_restore(lambda t: t, img, 128, accumulate='input'), 1e-6, 'test.py:65 as written returns the input')
    p, s, _ = D.psnr_ssim(ref, clean[:, :, :200, :248])
    assert abs(EV.psnr(out, clean[:, :, :200, :248].to(DEV)) - p) < 1e-3
    assert abs(EV.ssim(out, clean[:, :, :200, :248].to(DEV)) - s) < 1e-4


def test_ssim_vs_oracle():
    from fwair import evaluate as EV
    rs = np.random.RandomState(2)
    a = torch.from_numpy(rs.rand(3, 3, 70, 93).astype(np.float32) * 1.2 - 0.1)          # values outside [0, 1]: clipped first
    b = (a * 0.8 + torch.from_numpy(rs.rand(3, 3, 70, 93).astype(np.float32)) * 0.2)
    _, s, _ = D.psnr_ssim(a, b)
    assert abs(EV.ssim(a.to(DEV), b.to(DEV)) - s) < 2e-5
    assert abs(EV.ssim(a.clamp(0, 1).to(DEV), a.clamp(0, 1).to(DEV)) - 1.0) < 1e-6
