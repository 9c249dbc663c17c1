"""Shared test helpers: golden loading, seeded inputs (same recipe as tests/golden/make_golden.py)."""
import json
import os
import types

import numpy as np
import torch

import airnet_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

VARIANTS = {
    'all3': dict(degradation_embedding_method=['all_3_bands'], L=3, encoder_msa_type='freq'),
    'allDC': dict(degradation_embedding_method=['all_DC'], L=3, encoder_msa_type='freq'),
    'all2_L2': dict(degradation_embedding_method=['all_2_bands'], L=2, encoder_msa_type='freq'),
    'all3_origin': dict(degradation_embedding_method=['all_3_bands'], L=3, encoder_msa_type='origin'),
}


def load(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: (torch.from_numpy(np.asarray(z[k])) if z[k].dtype.kind != 'U' else z[k]) for k in z.files}


def schema(variant):
    with open(os.path.join(GOLDEN, 'schema.json')) as f:
        return [tuple(e) for e in json.load(f)[variant]]


def rnd(name, shape, scale=1.0):
    return O.seeded_tensor('input.' + name, shape) / 0.02 * scale


def synth_batch(B, size, tag):
    clean = torch.sigmoid(rnd(tag + 'clean', (B, 3, size, size), 1.5))
    q = (clean + rnd(tag + 'nq', clean.shape, 25 / 255.)).clamp(0, 1)
    k = (clean + rnd(tag + 'nk', clean.shape, 25 / 255.)).clamp(0, 1)
    return clean, q, k


def make_opt(variant, batch_size=2, **kw):
    d = dict(L=3, encoder_dim=256, encoder_embed_dim=28, embed_dim=56, batch_size=batch_size, patch_size=128,
             contrast_loss_weight=0.6, encoder_type='Uformer', decoder_type='Uformer', debug_mode=False,
             frequency_decompose_type='none', learnable_modulator=False, compute_dtype='fp32')
    d.update(VARIANTS[variant])
    d.update(kw)
    return types.SimpleNamespace(**d)


def close(a, b, tol, what=''):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, f'{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}'
    assert torch.isfinite(a).all(), f'{what}: non-finite values'
    scale = max(b.abs().max().item(), 1e-12)
    err = (a - b).abs().max().item() / scale
    assert err < tol, f'{what}: rel-to-max err {err:.3e} >= {tol:.1e} (scale {scale:.3e})'
    return err


def kdiff_state(variant='all3'):
    """Seeded state whose key encoder keeps its OWN name-seeded weights (golden model_all3_kdiff: the contrastive loss is O(1))."""
    sch = schema(variant)
    st = O.fill_state_seeded(sch)
    for name, shape, _ in sch:
        if name.startswith('E.E.encoder_k.') and O.is_parameter_key(name):
            st[name] = O.seeded_tensor(name, tuple(shape))
    return st


# golden model_all3_kdiff: gradient norms of the three 448 -> 65 536 head weights evaluated in FLOAT64 by the oracle (same graph, same
# weights).  dW = sum_t dY[t]^T xn[t] with sum_t dY[t] = 0 (BatchNorm backward) and xn[t] almost equal for all tokens (name-seeded
# weights): the sum is what survives a cancellation, and the reference's own f32 evaluation (torch CPU) is 0.24 .. 0.41 % away from
# the f64 value.  tests/test_oracle_model.py re-derives these constants; the HIP f32 path (exact-f32 MFMA products) must match THEM.
KDIFF_HEAD_WEIGHT_NORMS_F64 = {
    'E.E.encoder_q.mlp_head.0.1.weight': 0.028330916389931654,
    'E.E.encoder_q.mlp_head.1.1.weight': 0.03231436757293798,
    'E.E.encoder_q.mlp_head.2.1.weight': 0.015436637595669512,
}
