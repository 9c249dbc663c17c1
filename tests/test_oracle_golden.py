"""The CPU oracle (oracle/airnet_oracle.py) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py, run in the build container).  This is what pins the oracle.
Tolerances: fp32 CPU vs fp32 CPU on identical math -> 2e-5 relative to the tensor's max."""
import json
import os

import numpy as np
import pytest
import torch

import airnet_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def close(a, b, tol=2e-5, what=''):
    a, b = a.double(), b.double()
    scale = max(b.abs().max().item(), 1e-12)
    err = (a - b).abs().max().item() / scale
    assert err < tol, f'{what}: rel-to-max err {err:.3e} (scale {scale:.3e})'


def unit_state(prefix, g, extra_shapes):
    """Seeded state for a standalone unit: every grad key names a parameter; shapes from the grads."""
    st = {}
    for k, v in g.items():
        if k.startswith('g.'):
            st[k[2:]] = O.seeded_tensor(prefix + k[2:], v.shape).requires_grad_(True)
    for k, shp in extra_shapes.items():
        st[k] = O.seeded_tensor(prefix + k, shp).requires_grad_(True)
    return st


def check_grads(st, g, tol=5e-5):
    for k, v in g.items():
        if k.startswith('g.'):
            assert st[k[2:]].grad is not None, k
            close(st[k[2:]].grad, v, tol, k)


@pytest.mark.parametrize('n', [64, 128])
def test_frequency_decompose(n):
    g = load(f'unit_freq_decompose_{n}')
    x = g['x']
    for tag, ref in g.items():
        if tag == 'x':
            continue
        kind, size, inv = tag.split('|')
        inv = {'True': True, 'False': False}.get(inv, inv)
        out = O.frequency_decompose(x, kind, float(size), n, n, inv)
        assert out.shape == ref.shape, tag
        close(out, ref, 2e-5, tag)


def test_band_mask_bin_counts():
    """128^2/3 bands -> 1 / 6436 / 9947; 64^2 -> 1 / 1608 / 2487; 64^2 L=2 -> 1 / 4095."""
    assert [int(m.sum()) for m in O.band_masks('frequency_decompose_1', 0.5, 128, 128)] == [1, 6436, 9947]
    assert [int(m.sum()) for m in O.band_masks('frequency_decompose_1', 0.5, 64, 64)] == [1, 1608, 2487]
    assert [int(m.sum()) for m in O.band_masks('frequency_decompose_1', 1.0, 64, 64)] == [1, 4095]
    for m in O.band_masks('frequency_decompose_1', 0.5, 64, 64)[:2]:
        # Hermitian symmetry about the centre bin (needed for the real-filter formulation of LFS)
        mm = m[1:, 1:]
        assert torch.equal(mm, mm.flip(0, 1))


INTER = None


def inter_tensors():
    return tuple((O.seeded_tensor(f'input.inter{i}', (2, 64, 448)) / 0.02).requires_grad_(True) for i in range(3))


@pytest.mark.parametrize('dim,heads', [(56, 1), (112, 2)])
@pytest.mark.parametrize('method', ['all_3_bands', 'all_DC'])
@pytest.mark.parametrize('use_mask', [False, True])
def test_decoder_window_attention_lfs(dim, heads, method, use_mask):
    g = load(f'unit_wattn_{dim}_{method}_{"mask" if use_mask else "nomask"}')
    pre = f'unit_wattn_{dim}_{method}.'
    st = unit_state(pre, g, {})
    x = g['x'].clone().requires_grad_(True)
    it = inter_tensors()
    mask = O.shift_attn_mask(16, 16, 8, 4) if use_mask else None
    y = O.window_attention_lfs(st, '', x, heads, 4, it, mask, O.lfs_config([method]))
    close(y, g['y'], 2e-5, 'y')
    (y * g['dy']).sum().backward()
    close(x.grad, g['dx'], 5e-5, 'dx')
    for i in range(1, 3 if method == 'all_3_bands' else 2):
        close(it[i].grad, g[f'dinter{i}'], 5e-5, f'dinter{i}')
    check_grads(st, g)


@pytest.mark.parametrize('L', [3, 2])
@pytest.mark.parametrize('kind', ['intra', 'inter'])
@pytest.mark.parametrize('use_mask', [False, True])
def test_frequency_window_attention(L, kind, use_mask):
    g = load(f'unit_fwattn_{kind}_L{L}_{"mask" if use_mask else "nomask"}')
    st = unit_state(f'unit_fwattn_{kind}_L{L}.', g, {})
    x = g['x'].clone().requires_grad_(True)
    mask = O.shift_attn_mask(16, 16, 8, 4) if use_mask else None
    y = O.freq_window_attention(st, '', x, 2, L, kind, mask)
    close(y, g['y'], 2e-5, 'y')
    (y * g['dy']).sum().backward()
    close(x.grad, g['dx'], 5e-5, 'dx')
    check_grads(st, g)


def test_leff():
    g = load('unit_leff')
    st = unit_state('unit_leff.', g, {})
    x = g['x'].clone().requires_grad_(True)
    y = O.leff(st, '', x)
    close(y, g['y'], 2e-5, 'y')
    (y * g['dy']).sum().backward()
    close(x.grad, g['dx'], 5e-5, 'dx')
    check_grads(st, g)


@pytest.mark.parametrize('shift', [0, 4])
def test_decoder_block(shift):
    g = load(f'unit_decblock_s{shift}')
    st = unit_state(f'unit_decblock_s{shift}.', g, {})
    x = g['x'].clone().requires_grad_(True)
    it = tuple(t.detach() for t in inter_tensors())
    y = O.lewin_block_dec(st, '', x, 2, shift, it, O.lfs_config(['all_3_bands']))
    close(y, g['y'], 2e-5, 'y')
    (y * g['dy']).sum().backward()
    close(x.grad, g['dx'], 5e-5, 'dx')
    check_grads(st, g)


@pytest.mark.parametrize('msa', ['freq', 'origin'])
def test_encoder_block(msa):
    g = load(f'unit_encblock_{msa}')
    st = unit_state(f'unit_encblock_{msa}.', g, {})
    x = g['x'].clone().requires_grad_(True)
    y = O.lewin_block_enc(st, '', x, 2, 4, 3, msa)
    close(y, g['y'], 2e-5, 'y')
    (y * g['dy']).sum().backward()
    close(x.grad, g['dx'], 5e-5, 'dx')
    check_grads(st, g)
