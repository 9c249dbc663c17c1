"""The CPU oracle of the convolutional / ViT plug-ins (oracle/convnets_oracle.py) against goldens produced by the REAL reference
(tests/golden/make_golden.py convnets vit), plus the known-answer tests that anchor the deformable convolution, whose reference
line asserts (deform_conv.py:64; parity unpinned -- see the oracle's header)."""
import pytest
import torch
import torch.nn.functional as F

import airnet_oracle as O
import convnets_oracle as C
from helpers import close, load, make_opt, schema, synth_batch


def unit_state(prefix, g):
    st = {}
    for k, v in g.items():
        if k.startswith('g.'):
            st[k[2:]] = O.seeded_tensor(prefix + k[2:], v.shape).requires_grad_(True)
        elif k.startswith('s.'):
            st[k[2:]] = O.seeded_tensor(prefix + k[2:], v.shape)
    return st


def check_grads(st, g, tol=5e-5):
    n = 0
    for k, v in g.items():
        if k.startswith('g.'):
            close(st[k[2:]].grad, v, tol, k)
            n += 1
    assert n > 0


def bn_after(st, upd, key):
    m, v = upd[key]
    return st[key + 'running_mean'] * 0.9 + 0.1 * m, st[key + 'running_var'] * 0.9 + 0.1 * v


@pytest.mark.parametrize('tag,stride', [('s1', 1), ('s2', 2)])
def test_res_block(tag, stride):
    g = load(f'unit_resblock_{tag}')
    st = unit_state(f'unit_resblock_{tag}.', g)
    x = g['x'].clone().requires_grad_(True)
    upd = {}
    y = C.res_block(st, '', x, stride, True, upd)
    close(y, g['y'], 2e-5, 'y')
    (y * g['dy']).sum().backward()
    close(x.grad, g['dx'], 5e-5, 'dx')
    check_grads(st, g)
    for key in ('backbone.1.', 'backbone.4.', 'shortcut.1.'):
        m, v = bn_after(st, upd, key)
        close(m, g['s.' + key + 'running_mean'], 1e-5, key + 'running_mean')
        close(v, g['s.' + key + 'running_var'], 1e-5, key + 'running_var')


def test_sft_layer():
    g = load('unit_sft')
    st = unit_state('unit_sft.', g)
    x, it = g['x'].clone().requires_grad_(True), g['inter'].clone().requires_grad_(True)
    y = C.sft_layer(st, '', x, it)
    close(y, g['y'], 2e-5, 'y')
    (y * g['dy']).sum().backward()
    close(x.grad, g['dx'], 5e-5, 'dx')
    close(it.grad, g['dinter'], 5e-5, 'dinter')
    check_grads(st, g)


def seeded(prefix, variant):
    st = O.fill_state_seeded(schema(variant))
    return {k[len(prefix):]: v for k, v in st.items() if k.startswith(prefix)}


def test_resnet_encoder():
    g = load('model_resnet_encoder')
    st = seeded('E.E.encoder_q.', 'resnet_dgrn')
    for n in g['grad_names']:
        st[str(n)] = st[str(n)].clone().requires_grad_(True)
    x = g['x']
    with torch.no_grad():
        fea, out, inter = C.resnet_encoder(st, '', x, False)
    close(fea, g['fea_eval'], 2e-5, 'fea (eval)')
    close(out[0], g['out_eval'], 2e-5, 'out (eval)')
    close(inter, g['inter_eval'], 2e-5, 'inter (eval)')
    upd = {}
    fea, out, inter = C.resnet_encoder(st, '', x, True, upd)
    close(out[0], g['out_train'], 2e-5, 'out (train)')
    close(inter, g['inter_train'], 2e-5, 'inter (train)')
    ((out[0] * g['dout']).sum() + (inter * g['dinter']).sum()).backward()
    names = [str(n) for n in g['grad_names']]
    norms = torch.tensor([st[n].grad.norm().item() for n in names], dtype=torch.float64)
    close(norms, g['grad_norms'], 1e-4, 'gradient norms')
    check_grads(st, g, 1e-4)
    m, v = bn_after(st, upd, 'E.1.backbone.4.')
    close(v, g['s.E.1.backbone.4.running_var'], 1e-5, 'running_var')


def test_vit_encoder():
    g = load('model_vit_encoder')
    st = seeded('E.E.encoder_q.', 'vit_uformer')
    opt = make_opt('all3', encoder_type='ViT', encoder_dim=3)
    x = g['x']
    with torch.no_grad():
        fea, out, inter = C.vit_encoder(st, '', opt, x, False)
    close(fea, g['fea_eval'], 2e-5, 'fea (eval)')
    close(out[0], g['out_eval'], 2e-5, 'out (eval)')
    close(inter, g['inter_eval'], 2e-5, 'inter (eval)')
    names = [str(n) for n in g['grad_names']]
    for n in names:
        st[n] = st[n].clone().requires_grad_(True)
    fea, out, inter = C.vit_encoder(st, '', opt, x, True)
    close(out[0], g['out_train'], 2e-5, 'out (train)')
    close(inter, g['inter_train'], 2e-5, 'inter (train)')
    ((out[0] * g['dout']).sum() + (inter * g['dinter']).sum()).backward()
    norms = torch.tensor([st[n].grad.norm().item() for n in names], dtype=torch.float64)
    close(norms, g['grad_norms'], 2e-4, 'gradient norms')
    check_grads(st, g, 2e-4)


def _vit_case(gname, variant, size, tag, lamb_shape=None, **opt_kw):
    """Oracle ViT encoder against a reference golden: eval, and train mode with Dropout ON (the hashed masks both sides draw)."""
    import dropout_hash as DH
    from helpers import rnd
    g = load(gname)
    pre = 'E.E.encoder_q.'
    st = seeded(pre, variant)
    if lamb_shape is not None:
        for i in range(12):
            key = f'transformer.layers.{i}.0.fn.lamb'
            st[key] = O.seeded_tensor(pre + key, lamb_shape)
    opt = make_opt('all3', encoder_type='ViT', encoder_dim=3, **opt_kw)
    x = rnd(tag + 'x', (2, 3, size, size), 0.5)
    with torch.no_grad():
        fea, out, inter = C.vit_encoder(st, '', opt, x, False)
    close(fea, g['fea_eval'], 2e-5, 'fea (eval)')
    close(out[0], g['out_eval'], 2e-5, 'out (eval)')
    close(inter[:, :, ::4, ::4], g['inter_eval'], 2e-5, 'inter (eval)')
    names = [str(n) for n in g['grad_names']]
    for n in names:
        st[n] = st[n].clone().requires_grad_(True)
    upd = {}
    fea, out, inter = C.vit_encoder(st, '', opt, x, True, upd, drop=(int(g['drop_seed']), DH.site_base(pre), 0.1))
    close(out[0], g['out_train'], 5e-5, 'out (train, Dropout on)')
    close(inter[:, :, ::4, ::4], g['inter_train'], 5e-5, 'inter (train, Dropout on)')
    ((out[0] * rnd(tag + 'dout', out[0].shape)).sum() + (inter * rnd(tag + 'dinter', inter.shape)).sum()).backward()
    norms = torch.tensor([st[n].grad.norm().item() for n in names], dtype=torch.float64)
    close(norms, g['grad_norms'], 2e-4, 'gradient norms')
    check_grads(st, g, 2e-4)


def test_vit_encoder_256_with_dropout():
    """BASELINE configs[4]: N = 256 tokens (the reference class constructed with image_size=256)."""
    _vit_case('model_vit256_encoder', 'vit256_uformer', 256, 'vit256.')


@pytest.mark.parametrize('tag,ftype,bw,shape', [('3bands', '3_bands', False, (3, 1, 12)), ('DC', 'DC', False, (2, 1, 12)),
                                                ('DCbw', 'DC', True, (2, 2, 12))])
def test_vit_encoder_band_reweighting(tag, ftype, bw, shape):
    """encoder_ViT.py:51-66,85-92: learnable lamb per (band, [sample,] head) on the 64x64 attention maps."""
    _vit_case(f'model_vit_lamb_{tag}', 'vit_uformer', 128, 'vitlamb.', lamb_shape=shape, frequency_decompose_type=ftype,
              batch_wise_decompose=bw)


def test_dropout_hash_statistics():
    """The counter-based masks are Bernoulli(1 - p): rate, independence across sites and seeds, no short-range structure."""
    import numpy as np
    import dropout_hash as DH
    m = DH.keep_mask(1234, 7, (1 << 20,), 0.1)
    assert abs(m.mean() - 0.9) < 2e-3
    m2 = DH.keep_mask(1234, 8, (1 << 20,), 0.1)
    m3 = DH.keep_mask(1235, 7, (1 << 20,), 0.1)
    for o in (m2, m3, np.roll(m, 1), np.roll(m, 64)):
        assert abs(np.corrcoef(m, o)[0, 1]) < 5e-3
    assert DH.keep_mask(1, 2, (8, 8), 0.0).all()


# ---- DCNv2: known-answer tests ------------------------------------------------------------------------------------
def test_dcn_zero_offsets_is_half_a_convolution():
    """deform_conv.py:52-54 zero-initialises conv_offset_mask: offsets 0, mask = sigmoid(0) = 0.5 => DCN(x) = 0.5 * conv2d(x, W, pad 1)."""
    x, w = torch.randn(2, 8, 9, 11), torch.randn(6, 8, 3, 3)
    st = {'weight': w, 'conv_offset_mask.weight': torch.zeros(27, 16, 3, 3), 'conv_offset_mask.bias': torch.zeros(27)}
    y = C.dcn_layer(st, '', x, torch.randn(2, 8, 9, 11))
    close(y, 0.5 * F.conv2d(x, w, padding=1), 1e-5, 'zero offsets')


def test_dcn_integer_offsets_shift_the_taps():
    """Offsets (dy, dx) = (2, -1) on every tap, mask 1 => the plain 3x3 convolution of the image shifted by (2, -1), zero outside."""
    x, w = torch.randn(1, 5, 10, 12), torch.randn(4, 5, 3, 3)
    off = torch.zeros(1, 18, 10, 12)
    off[:, 0::2], off[:, 1::2] = 2.0, -1.0
    y = C.dcn_v2(x, off, torch.ones(1, 9, 10, 12), w)
    # tap (ky, kx) at output p reads x[p + (ky - 1 + 2, kx - 1 - 1)]: the zero-padded image, shifted up by 2 rows and right by 1 column
    xp = F.pad(x, (1 + 1, 1, 1, 1 + 2))                                  # (left, right, top, bottom) of the padded canvas
    ref = F.conv2d(xp, w)[:, :, 2:2 + 10, 0:12]
    close(y, ref, 1e-5, 'integer offsets')


def test_dcn_fractional_offset_interpolates_and_has_gradients():
    """A half-pixel offset on a linear ramp samples the ramp exactly (bilinear is exact on affine images); offsets, mask, input
    and weight all receive gradients (the chain the HIP backward is compared against)."""
    H = W = 8
    ramp = (torch.arange(W, dtype=torch.float32).view(1, 1, 1, W) + 10 * torch.arange(H, dtype=torch.float32).view(1, 1, H, 1)).contiguous()
    w = torch.zeros(1, 1, 3, 3); w[0, 0, 1, 1] = 1.0                      # centre tap only
    off = torch.zeros(1, 18, H, W); off[:, 8] = 0.5; off[:, 9] = 0.25     # tap 4 = centre: (dy, dx) = (0.5, 0.25)
    y = C.dcn_v2(ramp, off, torch.ones(1, 9, H, W), w)
    close(y[0, 0, 1:6, 1:6], ramp[0, 0, 1:6, 1:6] + 5.0 + 0.25, 1e-5, 'affine image')
    x = torch.randn(1, 3, H, W, requires_grad=True)
    off = (torch.randn(1, 18, H, W) * 0.7).requires_grad_(True)
    m = torch.rand(1, 9, H, W, requires_grad=True)
    ww = torch.randn(2, 3, 3, 3, requires_grad=True)
    C.dcn_v2(x, off, m, ww).square().sum().backward()
    assert all(float(t.grad.abs().sum()) > 0 for t in (x, off, m, ww))
