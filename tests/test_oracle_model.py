"""Whole-model oracle (oracle/airnet_oracle.py) against goldens produced by the REAL reference: eval output,
PSNR, train-mode loss / logits / restored, per-parameter gradient norms, MoCo queue and BN running stats."""
import pytest
import torch

import airnet_oracle as O
from helpers import VARIANTS, close, load, make_opt, schema, synth_batch


@pytest.mark.parametrize('variant', list(VARIANTS))
def test_eval_forward(variant):
    g = load(f'model_{variant}')
    st = O.fill_state_seeded(schema(variant))
    opt = make_opt(variant)
    clean, q, k = synth_batch(2, 128, 'model.')
    with torch.no_grad():
        out = O.airnet_forward(st, opt, q, q, False)
    close(out, g['restored_eval'], 1e-4, 'restored_eval')
    assert abs(O.psnr(out, clean) - float(g['psnr_eval'])) < 1e-3


@pytest.mark.parametrize('variant', ['all3', 'all2_L2'])
def test_train_step(variant):
    g = load(f'model_{variant}')
    st = O.fill_state_seeded(schema(variant))
    names = [str(n) for n in g['grad_names']]
    for n in names:
        st[n] = st[n].clone().requires_grad_(True)
    opt = make_opt(variant)
    clean, q, k = synth_batch(2, 128, 'model.')
    restored, logits, labels = O.airnet_forward(st, opt, q, k, True)
    loss, l1, contrast = O.training_loss(opt, restored, logits, labels, clean)
    close(restored, g['restored_train'], 1e-4, 'restored_train')
    close(torch.stack(logits), g['logits'], 1e-4, 'logits')
    close(loss, g['loss'], 1e-5, 'loss')
    loss.backward()
    norms = torch.tensor([st[n].grad.norm().item() for n in names])
    close(norms, g['grad_norms'], 2e-3, 'per-parameter grad norms')
    for key, val in g.items():
        if key.startswith('g.'):
            close(st[key[2:]].grad, val, 2e-3, key)
    close(st['E.E.queue'], g['queue_after'], 1e-5, 'queue')
    assert int(st['E.E.queue_ptr']) == int(g['queue_ptr_after'])
    close(st['E.E.encoder_q.norm.0.0.running_mean'], g['bn_q0_running_mean'], 1e-4, 'bn running mean (q)')
    close(st['E.E.encoder_q.norm.0.0.running_var'], g['bn_q0_running_var'], 1e-4, 'bn running var (q)')
    close(st['E.E.encoder_k.norm.0.0.running_mean'], g['bn_k0_running_mean'], 1e-4, 'bn running mean (k)')


def test_train_step_with_order_one_contrastive_loss():
    """golden model_all3_kdiff (reference run with an independently seeded key encoder): contrast ~ 1.27, so the InfoNCE path and the
    query encoder's backward are pinned at real magnitudes, not at the 6e-6 / 1e-7 of the q == k goldens."""
    from helpers import kdiff_state
    g = load('model_all3_kdiff')
    st = kdiff_state()
    names = [str(n) for n in g['grad_names']]
    for n in names:
        st[n] = st[n].clone().requires_grad_(True)
    opt = make_opt('all3')
    clean, q, k = synth_batch(2, 128, 'model.')
    restored, logits, labels = O.airnet_forward(st, opt, q, k, True)
    loss, l1, contrast = O.training_loss(opt, restored, logits, labels, clean)
    assert float(g['contrast']) > 1.0
    close(contrast, g['contrast'], 1e-5, 'contrast')
    close(torch.stack(logits), g['logits'], 1e-4, 'logits')
    close(loss, g['loss'], 1e-5, 'loss')
    loss.backward()
    norms = torch.tensor([st[n].grad.norm().item() for n in names])
    close(norms, g['grad_norms'], 1e-3, 'per-parameter grad norms')
    enc = [i for i, n in enumerate(names) if n.startswith('E.E.encoder_q.')]
    rel = ((norms[enc] - g['grad_norms'][enc]).abs() / g['grad_norms'][enc].clamp_min(1e-12))
    assert rel.max() < 2e-2 and rel.median() < 1e-3, f'encoder grad norms: max rel {rel.max():.2e} ({names[enc[int(rel.argmax())]]})'
    for key, val in g.items():
        if key.startswith('g.'):
            close(st[key[2:]].grad, val, 2e-3, key)
    close(st['E.E.queue'], g['queue_after'], 1e-5, 'queue')


def test_kdiff_head_weight_norms_in_float64():
    """The float64 evaluation of the kdiff graph: the three head-weight gradient norms the GPU test judges the HIP path against
    (helpers.KDIFF_HEAD_WEIGHT_NORMS_F64), and the distance of the reference's f32 values from them (0.24 .. 0.41 %)."""
    from helpers import KDIFF_HEAD_WEIGHT_NORMS_F64 as F64, kdiff_state
    g = load('model_all3_kdiff')
    st = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kdiff_state().items()}
    names = [str(n) for n in g['grad_names']]
    for n in names:
        st[n] = st[n].clone().requires_grad_(True)
    clean, q, k = (t.double() for t in synth_batch(2, 128, 'model.'))
    restored, logits, labels = O.airnet_forward(st, make_opt('all3'), q, k, True)
    O.training_loss(make_opt('all3'), restored, logits, labels, clean)[0].backward()
    for n, want in F64.items():
        got = float(st[n].grad.norm())
        assert abs(got - want) < 1e-9 * want, f'{n}: {got!r}'
        ref = float(g['grad_norms'][names.index(n)])
        assert 2e-3 < abs(ref - want) / want < 5e-3


def test_256_eval_and_train_step():
    """Resolution-generic construction: the reference classes built with img_size=256 (golden model256_all3) -- the
    bottleneck is 16x16 there, so its odd blocks shift, the LFS heads average 256 tokens and the band split is a 256-point DFT."""
    g = load('model256_all3')
    st = O.fill_state_seeded(schema('all3'))
    st['E.E.queue'] = torch.nn.functional.normalize(O.seeded_tensor('E.E.queue', (3, 256, 3)) / 0.02, dim=1)   # K = 3 * batch_size
    opt = make_opt('all3', batch_size=1, patch_size=256)
    clean, q, k = synth_batch(1, 256, 'model256.')
    with torch.no_grad():
        out = O.airnet_forward(st, opt, q, q, False)
    close(out, g['restored_eval'], 1e-4, 'restored_eval')
    assert abs(O.psnr(out, clean) - float(g['psnr_eval'])) < 1e-3
    names = [str(n) for n in g['grad_names']]
    for n in names:
        st[n] = st[n].clone().requires_grad_(True)
    restored, logits, labels = O.airnet_forward(st, opt, q, k, True)
    loss, l1, contrast = O.training_loss(opt, restored, logits, labels, clean)
    close(restored, g['restored_train'], 1e-4, 'restored_train')
    close(torch.stack(logits), g['logits'], 1e-4, 'logits')
    close(loss, g['loss'], 1e-5, 'loss')
    loss.backward()
    norms = torch.tensor([st[n].grad.norm().item() for n in names])
    close(norms, g['grad_norms'], 2e-3, 'per-parameter grad norms')
    for key, val in g.items():
        if key.startswith('g.'):
            close(st[key[2:]].grad, val, 2e-3, key)


def test_debug_mode_payload():
    """`opt.debug_mode` (plot_MSA_frequency.py:47, plot_embed_lamb_curve.py:48): (restored, visual_freqs) with one
    [spectrum_before, spectrum_after, embed_lamb] per block (decoder_Uformer.py:668-673,731-736,753-756,1168-1169)."""
    g = load('debug_all3')
    st = O.fill_state_seeded(schema('all3'))
    opt = make_opt('all3', debug_mode=True)
    clean, q, k = synth_batch(2, 128, 'model.')
    with torch.no_grad():
        restored, vf = O.airnet_forward(st, opt, q, q, False)
    close(restored, g['restored'], 1e-4, 'restored')
    assert [len(layer) for layer in vf] == g['layers'].tolist() == [2, 2, 8, 8, 2, 2, 8, 8, 2, 2]
    for li, layer in enumerate(vf):
        for bi, (before, after, lamb) in enumerate(layer):
            close(before, g[f'before.{li}.{bi}'], 1e-4, f'spectrum before {li}.{bi}')
            close(after, g[f'after.{li}.{bi}'], 1e-4, f'spectrum after {li}.{bi}')
            close(lamb, g[f'lamb.{li}.{bi}'], 1e-4, f'embed_lamb {li}.{bi}')
