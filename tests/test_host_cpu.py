"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/fwair.h declares,
the module tree reproduces the reference's state_dict schema, options keep the reference's names / defaults,
and the product path refuses to run without the HIP device (no CPU fallback, no oracle import)."""
import math
import os
import re
import subprocess
import sys

import pytest
import torch

from helpers import VARIANTS, make_opt, schema

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd')


def test_library_exports_every_declared_symbol():
    from fwair import lib
    protos = lib.protos()
    hdr = open(lib.HEADER_PATH).read()
    declared = set(re.findall(r'\bint\s+(fw_\w+)\s*\(', hdr))
    assert declared == set(protos), declared ^ set(protos)
    assert len(declared) >= 40
    nm = subprocess.run(['nm', '-D', '--defined-only', lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (fw_\w+)', nm))
    assert declared <= exported, declared - exported
    assert lib.lib().fw_attn_lfs_table_elems() == 29696
    # the package's own copy of the header (made by build.sh, what lib.py parses) is the repository's include/fwair.h
    own = os.path.join(PKG, 'fwair', 'fwair.h')
    if os.path.exists(own):
        assert open(own).read() == open(os.path.join(ROOT, 'include', 'fwair.h')).read(), 'stale fwair/fwair.h: re-run build.sh'


@pytest.mark.parametrize('variant', list(VARIANTS))
def test_state_dict_schema_matches_reference(variant):
    from net.model import AirNet
    net = AirNet(make_opt(variant))
    mine = [(k, list(v.shape), str(v.dtype).replace('torch.', '')) for k, v in net.state_dict().items()]
    ref = [(k, list(s), d) for k, s, d in schema(variant)]
    assert mine == ref
    # key encoder starts as a copy of the query encoder and is frozen (moco.py:33-35)
    for pq, pk in zip(net.E.E.encoder_q.parameters(), net.E.E.encoder_k.parameters()):
        assert torch.equal(pq, pk) and not pk.requires_grad
    mf = net.E.E.encoder_q.uformer.encoderlayer_0.blocks[0].attn_inter.mask_freq if variant != 'all3_origin' else None
    if mf is not None:
        L = make_opt(variant).L
        assert mf.shape == (1, 1, 64 * L, 64 * L) and float(mf[0, 0, 0, 0]) == -100.0 and float(mf[0, 0, 0, 64]) == 0.0


def test_no_cpu_fallback_and_no_oracle_import():
    from fwair import lib
    x = torch.zeros(8, 8)
    with pytest.raises(RuntimeError, match='not on a HIP device'):
        lib.call('fw_fill', x, 64, 0.0)                      # a CPU tensor is rejected before any launch
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(import|from)\s+(airnet_oracle|oracle|refshim)', src, re.M), f
                assert '/root/reference' not in src, f     # nothing may read the reference at run time


def test_unsupported_configurations_fail_loudly():
    from net.model import AirNet
    with pytest.raises(NotImplementedError):
        AirNet(make_opt('all3', degradation_embedding_method=['residual']))        # the CLI default does not run in the reference either
    with pytest.raises(NotImplementedError):                                        # ViT: N = (S/16)^2 keys per head must be 64 or 256
        AirNet(make_opt('all3', encoder_type='ViT', encoder_dim=3, out_channels=3, batch_wise_decompose=False,
                        degradation_embedding_method=['None'], patch_size=64))
    net = AirNet(make_opt('all3', encoder_type='ViT', encoder_dim=3, out_channels=3, batch_wise_decompose=False,
                          degradation_embedding_method=['None'], frequency_decompose_type='3_bands'))
    lamb = net.E.E.encoder_q.transformer.layers[0][0].fn.lamb                       # the band re-weighting of the ViT attention (encoder_ViT.py:55-63)
    assert tuple(lamb.shape) == (3, 1, 12) and lamb.requires_grad           # [bands, 1 (not batch-wise), heads]


@pytest.mark.parametrize('variant,kw', [
    ('resnet_dgrn', dict(encoder_type='ResNet', decoder_type='ResNet', encoder_dim=256, patch_size=64, degradation_embedding_method=['residual'])),
    ('vit_uformer', dict(encoder_type='ViT', decoder_type='Uformer', encoder_dim=3, degradation_embedding_method=['None'], out_channels=3,
                         batch_wise_decompose=False))])
def test_convnet_and_vit_seam_schemas_match_reference(variant, kw):
    """BASELINE configs[0] / [4]: the ResNet / DGRN / ViT names of the seam (net/model.py:3,17,31) build, with the reference's
    state_dict keys, shapes, dtypes and order (tests/golden/schema.json, dumped from the reference's own AirNet(opt))."""
    from net.model import AirNet
    net = AirNet(make_opt('all3', **kw))
    mine = [(k, list(v.shape), str(v.dtype).replace('torch.', '')) for k, v in net.state_dict().items()]
    assert mine == [(k, list(s), d) for k, s, d in schema(variant)]
    for pq, pk in zip(net.E.E.encoder_q.parameters(), net.E.E.encoder_k.parameters()):
        assert torch.equal(pq, pk) and not pk.requires_grad
    if variant == 'resnet_dgrn':                             # deform_conv.py:52-54: offsets / masks start at zero
        d = net.R.R.body[0].body[0].dgm1.dcn
        assert float(d.conv_offset_mask.weight.abs().max()) == 0 and d.bias is None and tuple(d.weight.shape) == (64, 64, 3, 3)


def test_patch_size_sets_the_model_resolution():
    """`opt.patch_size` reaches encoder / decoder `img_size`.  At 256 the bottleneck is 16x16, so its odd blocks
    shift (decoder_Uformer.py:531-533 only zeroes the shift when the map is one window); the state dict does not change."""
    from net.model import AirNet
    with pytest.raises(NotImplementedError):
        AirNet(make_opt('all3', patch_size=64))           # 4x4 bottleneck under an 8x8 window: fails in the reference too
    n128, n256 = AirNet(make_opt('all3')), AirNet(make_opt('all3', patch_size=256))
    assert n128.R.R.bottleneck_0.blocks[1].shift_size == 0 and n256.R.R.bottleneck_0.blocks[1].shift_size == 4
    assert n128.E.E.encoder_q.uformer.conv.blocks[1].shift_size == 0 and n256.E.E.encoder_q.uformer.conv.blocks[1].shift_size == 4
    assert [(k, tuple(v.shape)) for k, v in n128.state_dict().items()] == [(k, tuple(v.shape)) for k, v in n256.state_dict().items()]


def test_flat_layout_packs_frequency_attention_tables():
    """engine._layout: every tensor starts 16-byte aligned, except the 2nd..9th relative-position table of a
    FrequencyWindowAttention, which continue the first so that the kernel operand [L*L, 225, heads] is one dense view."""
    from net.model import AirNet
    from fwair import engine as E
    from fwair.modules import FrequencyWindowAttention
    enc = AirNet(make_opt('all3')).E.E.encoder_q
    params = E.ordered_parameters(enc)
    offs, total = E._layout(params)
    where = {id(p): o for p, o in zip(params, offs)}
    assert len(where) == len(list(enc.parameters())) and total % 8 == 0
    spans = sorted((o, o + p.numel()) for p, o in zip(params, offs))
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))                   # no overlap
    n = 0
    for mod in enc.modules():
        if isinstance(mod, FrequencyWindowAttention):
            tabs = list(mod.relative_position_bias_table)
            assert where[id(tabs[0])] % 8 == 0
            assert [where[id(t)] for t in tabs] == [where[id(tabs[0])] + i * tabs[0].numel() for i in range(len(tabs))]
            n += 1
    assert n == 20
    unpacked = [p for p in params if not getattr(p, '_fw_pack', False)]
    assert all(where[id(p)] % 8 == 0 for p in unpacked)


def test_synthetic_rain_haze_and_mixed_task_batches():
    """The stand-ins for the rain / haze pairs the reference only reads from disk and the per-sample task cycle of
    dataset_utils.py:99 (BASELINE configs[2])."""
    from fwair import augment as A
    from fwair.synthetic import synth_task_batch
    g = torch.Generator(); g.manual_seed(3)
    clean = (torch.rand(3, 64, 64, generator=g) * 255).to(torch.uint8)
    rain = A.add_rain(clean, g)
    assert rain.dtype == torch.uint8 and bool((rain >= clean).all()) and 0.002 < float((rain > clean).float().mean()) < 0.5
    lit = (rain[0].int() - clean[0].int()) > 0
    ys, xs = torch.nonzero(lit, as_tuple=True)                     # streaks run along one diagonal: rows and columns co-vary
    assert abs(float(torch.corrcoef(torch.stack([ys.float(), xs.float()]))[0, 1])) < 1.0
    haze = A.add_haze(clean, beta=1.0)
    t_top, t_bot = math.exp(-1.0), math.exp(-0.1)
    exp_top = (clean[:, 0].float() / 255 * t_top + 0.8 * (1 - t_top)) * 255
    assert float((haze[:, 0].float() - exp_top).abs().max()) <= 1.0
    exp_bot = (clean[:, -1].float() / 255 * t_bot + 0.8 * (1 - t_bot)) * 255
    assert float((haze[:, -1].float() - exp_bot).abs().max()) <= 1.0
    assert A.degrade(clean, 'denoising_0', g).shape == clean.shape
    with pytest.raises(ValueError):
        A.degrade(clean, 'deblurring', g)
    tasks = ['denoising_15', 'denoising_25', 'denoising_50', 'deraining', 'dehazing']
    c, d1, d2 = synth_task_batch(5, 32, tasks, 7, 'cpu')
    c_, d1_, _ = synth_task_batch(5, 32, tasks, 7, 'cpu')
    assert c.shape == d1.shape == d2.shape == (5, 3, 32, 32) and torch.equal(d1, d1_) and torch.equal(c, c_)
    assert float((d1[3] - c[3]).min()) >= 0 and float((d1[3] - c[3]).max()) > 0          # rain only brightens
    assert float((d1[0] - c[0]).abs().mean()) > float((d1[4] - c[4]).abs().mean()) * 0 and 0 <= float(d1.min()) and float(d1.max()) <= 1
    b = A.training_batch([clean, clean], 16, ['deraining', 'denoising_25'], g)
    assert b[0].shape == (2, 3, 16, 16) and float((b[0][0] - b[2][0]).min()) >= 0


def test_c_abi_header_is_plain_c():
    """include/fwair.h is the drop-in boundary: it must compile as C99 (no torch / C++ types in the signatures) and as C++."""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, 'include', 'fwair.h')
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only', '-x', 'c', hdr], check=True)
    subprocess.run(['g++', '-std=c++17', '-fsyntax-only', '-x', 'c++', hdr], check=True)


def test_option_defaults(monkeypatch):
    monkeypatch.setattr(sys, 'argv', ['x', '--degradation_embedding_method', 'all_3_bands', '--de_type', 'denoising_25', 'denoising_25'])
    sys.modules.pop('option', None)
    from option import options as o
    assert (o.batch_size, o.lr, o.encoder_dim, o.L, o.patch_size, o.encoder_msa_type) == (2, 2e-4, 256, 3, 128, 'freq')
    assert o.contrast_loss_weight is None and o.default_contrast_loss_weight == 0.6
    assert o.ckpt_path == 'output/tmp/ckpt/' and o.compute_dtype == 'fp32'
    sys.modules.pop('option', None)


def test_band_masks_equal_oracle():
    import airnet_oracle as O
    from fwair import lfs
    for kind, size, n in (('frequency_decompose_1', 0.5, 128), ('frequency_decompose_1', 1.0, 64), ('frequency_decompose', 1 / 3., 64)):
        for a, b in zip(lfs.band_masks_shifted(kind, size, n, n), O.band_masks(kind, size, n, n)):
            assert torch.equal(a, b)


def test_droppath_plan_and_tile_grid():
    """Host logic without a GPU: the one-draw-per-step DropPath plan (record -> ready -> dirty on deviation -> record) and the
    tile grid of the device-side evaluation (test.py:47-48)."""
    import torch
    from fwair import functional as Fn
    from fwair.evaluate import tile_origins
    assert tile_origins(200, 128) == [0, 72] and tile_origins(128, 128) == [0] and tile_origins(264, 128) == [0, 128, 136]
    p = Fn._DropPathPool()
    states = []
    for _ in range(3):
        p.begin('cpu')
        a, b = p.draw(4, 0.9, 'cpu'), p.draw(6, 0.8, 'cpu')
        assert a.shape == (4,) and b.shape == (6,)
        assert set((a * 0.9).round().tolist()) <= {0.0, 1.0} and set((b * 0.8).round().tolist()) <= {0.0, 1.0}
        states.append(p.state)
    assert states == ['record', 'ready', 'ready'] and p.plan == [(4, 0.9), (6, 0.8)]
    p.begin('cpu')
    p.draw(4, 0.9, 'cpu')
    assert p.draw(5, 0.8, 'cpu').shape == (5,) and p.state == 'dirty'          # another batch size: per-call draws for this step
    p.begin('cpu')
    assert p.state == 'record' and p.plan == []


def test_device_input_pipeline_semantics():
    """fwair/augment.py against the reference's numpy semantics (image_utils.py:133-160 on HWC arrays, dataset_utils.py:126)."""
    import numpy as np
    import torch
    from fwair import augment as A
    rs = np.random.RandomState(0)
    hwc = rs.randint(0, 256, (24, 40, 3)).astype(np.uint8)

    def ref(img, mode):
        k, flip = mode // 2, mode % 2 == 1
        out = np.rot90(img, k=k) if k else img
        return np.flipud(out) if flip else out

    chw = torch.from_numpy(hwc).permute(2, 0, 1)
    for mode in range(8):
        got = A.augment(chw, mode).permute(1, 2, 0).numpy()
        assert np.array_equal(got, ref(hwc, mode)), f'mode {mode}'
    g = torch.Generator().manual_seed(5)
    noisy = A.add_noise(chw, 25, g)
    assert noisy.dtype == torch.uint8 and noisy.shape == chw.shape
    d = noisy.float() - chw.float()
    assert 15 < float(d.std()) < 30                                       # sigma 25 on the uint8 grid, clipped at the ends
    d1, d2, c1, c2 = A.training_pair(chw, 16, sigma=15, generator=g)
    assert d1.shape == d2.shape == c1.shape == c2.shape == (3, 16, 16) and float(c1.max()) <= 1.0
    b = A.training_batch([chw, chw], 16, [15, 50], g)
    assert len(b) == 4 and b[0].shape == (2, 3, 16, 16)
