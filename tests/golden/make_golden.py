"""Golden-vector generator -- runs ONLY in the build container, where /root/reference exists.

Imports the unmodified reference through ``oracle/refshim.py`` (timm stand-in, no-op
``.cuda()``), fills every floating tensor from the name-seeded RNG of
``oracle/airnet_oracle.py`` (weights are never stored), runs the reference on seeded
inputs with DropPath neutralised and stores inputs + outputs as small ``.npz`` files
(``numpy.load`` with ``allow_pickle=False`` reads them) plus the state-dict schema as JSON.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [unit] [model] [moco] [model256] [debug] [convnets] [vit] [vit256] [vitlamb] [kdiff]

A fixture is data (inputs / expected outputs); no reference source text is stored.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import refshim                     # noqa: E402
import airnet_oracle as O          # noqa: E402

BASE_ARGS = ['--contrast_loss_weight', '0.6', '--degradation_embedding_method', 'all_3_bands',
             '--de_type', 'denoising_25', 'denoising_25']
_ARGV = sys.argv[1:]
opt = refshim.install(BASE_ARGS)

from net.model import AirNet                                   # noqa: E402
from net import decoder_Uformer as RD                          # noqa: E402
from net import encoder_Uformer as RE                          # noqa: E402
from net.utils.frequency_decompose import FrequencyDecompose   # noqa: E402
from net.utils.leff import LeFF                                # noqa: E402

torch.set_num_threads(8)


def rnd(name, shape, scale=1.0):
    return O.seeded_tensor('input.' + name, shape) / 0.02 * scale


def seed_module(mod, prefix):
    """Fill a reference module from the name-seeded RNG (keys = prefix + state_dict key)."""
    sd = mod.state_dict()
    new = {}
    for k, v in sd.items():
        tail = k.rsplit('.', 1)[-1]
        if not v.is_floating_point() or tail == 'mask_freq':
            new[k] = v
        elif tail == 'queue':
            new[k] = torch.nn.functional.normalize(O.seeded_tensor(prefix + k, v.shape) / 0.02, dim=1)
        else:
            new[k] = O.seeded_tensor(prefix + k, v.shape)
    mod.load_state_dict(new)
    for m in mod.modules():
        if hasattr(m, 'drop_prob'):
            m.drop_prob = 0.0
    return mod


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', name, '%.1f KB' % (os.path.getsize(path) / 1024))


def grads_of(mod, loss):
    for p in mod.parameters():
        p.grad = None
    loss.backward()
    return {k: p.grad for k, p in mod.named_parameters() if p.grad is not None}


# ---------------------------------------------------------------------------------------
def gen_unit():
    # (i) FrequencyDecompose, all three kinds, inverse True / False / 'visual'
    for n in (64, 128):
        x = rnd(f'fd{n}', (1, 2, n, n))
        out = {'x': x}
        for kind, size in (('frequency_decompose', 1 / 3.), ('frequency_decompose_1', 0.5),
                           ('frequency_decompose_dc', 0.5), ('frequency_decompose', 1.0)):
            for inv in (True, False, 'visual'):
                if kind == 'frequency_decompose_dc' and inv is not True:
                    continue
                fd = FrequencyDecompose(kind, size, n, n, inverse=inv)
                tag = f'{kind}|{size:.4f}|{inv}'
                out[tag] = fd(x)
        save(f'unit_freq_decompose_{n}', **out)

    inter = tuple(rnd(f'inter{i}', (2, 64, 448)) for i in range(3))

    # (ii) decoder WindowAttention with LFS: dims 56/h1, 112/h2; mask on/off; 3 bands / DC
    for dim, heads in ((56, 1), (112, 2)):
        for method in ('all_3_bands', 'all_DC'):
            pre = f'unit_wattn_{dim}_{method}.'
            wa = seed_module(RD.WindowAttention((16, 16), dim, (8, 8), heads,
                                                all_degradation_embedding_method=[method]), pre)
            for use_mask in (False, True):
                x = rnd(pre + 'x', (8, 64, dim)).requires_grad_(True)
                it = tuple(t.clone().requires_grad_(True) for t in inter)
                mask = O.shift_attn_mask(16, 16, 8, 4) if use_mask else None
                y, _ = wa(x, all_inter=it, mask=mask)
                w = rnd(pre + 'dy', y.shape)
                g = grads_of(wa, (y * w).sum())
                arrs = {'x': x, 'y': y, 'dy': w, 'dx': x.grad}
                for i in range(1, 3 if method == 'all_3_bands' else 2):
                    arrs[f'dinter{i}'] = it[i].grad
                for k, v in g.items():
                    arrs['g.' + k] = v
                save(f'unit_wattn_{dim}_{method}_{"mask" if use_mask else "nomask"}', **arrs)

    # (iii) FrequencyWindowAttention intra / inter (L=3 and L=2), with shift mask
    for L in (3, 2):
        for kind in ('intra', 'inter'):
            pre = f'unit_fwattn_{kind}_L{L}.'
            fa = seed_module(RE.FrequencyWindowAttention(56, (8, 8), 2, type=kind, L=L), pre)
            for use_mask in (False, True):
                x = rnd(pre + 'x', (L * 1 * 4, 64, 56)).requires_grad_(True)
                mask = O.shift_attn_mask(16, 16, 8, 4) if use_mask else None
                y, _, _ = fa(x, mask=mask)
                w = rnd(pre + 'dy', y.shape)
                g = grads_of(fa, (y * w).sum())
                arrs = {'x': x, 'y': y, 'dy': w, 'dx': x.grad}
                for k, v in g.items():
                    arrs['g.' + k] = v
                save(f'unit_fwattn_{kind}_L{L}_{"mask" if use_mask else "nomask"}', **arrs)

    # (iv) LeFF
    pre = 'unit_leff.'
    lf = seed_module(LeFF(56, 224), pre)
    x = rnd(pre + 'x', (2, 256, 56)).requires_grad_(True)
    y = lf(x)
    w = rnd(pre + 'dy', y.shape)
    g = grads_of(lf, (y * w).sum())
    arrs = {'x': x, 'y': y, 'dy': w, 'dx': x.grad}
    arrs.update({'g.' + k: v for k, v in g.items()})
    save('unit_leff', **arrs)

    # (v) LeWin blocks: decoder (shift 0 / 4), encoder freq (shift 4), encoder origin
    for shift in (0, 4):
        pre = f'unit_decblock_s{shift}.'
        blk = seed_module(RD.LeWinTransformerBlock(112, (16, 16), 2, win_size=8, shift_size=shift,
                                                   all_degradation_embedding_method=['all_3_bands']), pre)
        x = rnd(pre + 'x', (2, 256, 112)).requires_grad_(True)
        y = blk(x, all_inter=inter)
        w = rnd(pre + 'dy', y.shape)
        g = grads_of(blk, (y * w).sum())
        arrs = {'x': x, 'y': y, 'dy': w, 'dx': x.grad}
        arrs.update({'g.' + k: v for k, v in g.items()})
        save(f'unit_decblock_s{shift}', **arrs)
    for msa in ('freq', 'origin'):
        pre = f'unit_encblock_{msa}.'
        blk = seed_module(RE.LeWinTransformerBlock(56, (16, 16), 2, win_size=8, shift_size=4,
                                                   encoder_msa_type=msa, L=3), pre)
        x = rnd(pre + 'x', (3 * 2, 256, 56)).requires_grad_(True)
        y, _, _ = blk(x)
        w = rnd(pre + 'dy', y.shape)
        g = grads_of(blk, (y * w).sum())
        arrs = {'x': x, 'y': y, 'dy': w, 'dx': x.grad}
        arrs.update({'g.' + k: v for k, v in g.items()})
        save(f'unit_encblock_{msa}', **arrs)


def set_opt(**kw):
    for k, v in kw.items():
        setattr(opt, k, v)


def schema_of(net):
    return [(k, list(v.shape), str(v.dtype).replace('torch.', '')) for k, v in net.state_dict().items()]


def synth_batch(B, size, tag):
    clean = torch.sigmoid(rnd(tag + 'clean', (B, 3, size, size), 1.5))
    q = (clean + rnd(tag + 'nq', clean.shape, 25 / 255.)).clamp(0, 1)
    k = (clean + rnd(tag + 'nk', clean.shape, 25 / 255.)).clamp(0, 1)
    return clean, q, k


def gen_model():
    variants = {
        'all3': dict(degradation_embedding_method=['all_3_bands'], L=3, encoder_msa_type='freq'),
        'allDC': dict(degradation_embedding_method=['all_DC'], L=3, encoder_msa_type='freq'),
        'all2_L2': dict(degradation_embedding_method=['all_2_bands'], L=2, encoder_msa_type='freq'),
        'all3_origin': dict(degradation_embedding_method=['all_3_bands'], L=3, encoder_msa_type='origin'),
    }
    schemas = {}
    for name, kw in variants.items():
        set_opt(batch_size=2, **kw)
        t0 = time.time()
        net = seed_module(AirNet(opt), '')
        for pq, pk in zip(net.E.E.encoder_q.parameters(), net.E.E.encoder_k.parameters()):
            pk.data.copy_(pq.data)
        schemas[name] = schema_of(net)
        clean, q, k = synth_batch(2, 128, 'model.')
        net.eval()
        with torch.no_grad():
            restored_eval = net(x_query=q, x_key=q)
        arrs = {'restored_eval': restored_eval,
                'psnr_eval': O.psnr(restored_eval, clean), 'psnr_input': O.psnr(q, clean)}
        if name in ('all3', 'all2_L2'):
            net.train()
            restored, logits, labels = net(x_query=q, x_key=k)
            CE = torch.nn.CrossEntropyLoss()
            contrast = sum(CE(logits[i], labels[i]) for i in range(opt.L)) / opt.L
            l1 = torch.nn.L1Loss()(restored, clean)
            loss = l1 + opt.contrast_loss_weight * contrast
            g = grads_of(net, loss)
            names = sorted(g.keys())
            arrs.update({'restored_train': restored, 'logits': torch.stack(logits, 0),
                         'loss': loss, 'l1': l1, 'contrast': contrast,
                         'grad_names': np.array(names), 'grad_norms': np.array([g[n].norm().item() for n in names]),
                         'queue_after': net.E.E.queue, 'queue_ptr_after': net.E.E.queue_ptr})
            for n in names:
                if g[n].numel() <= 4096 and ('blocks.0.' in n or 'mlp.' in n):
                    arrs['g.' + n] = g[n]
            for n in ('R.R.output_proj.proj.0.weight', 'R.R.input_proj.proj.0.weight',
                      'E.E.encoder_q.uformer.input_proj.proj.0.weight'):
                arrs['g.' + n] = g[n]
            bn = net.E.E.encoder_q.norm[0][0]
            arrs['bn_q0_running_mean'] = bn.running_mean
            arrs['bn_q0_running_var'] = bn.running_var
            arrs['bn_k0_running_mean'] = net.E.E.encoder_k.norm[0][0].running_mean
        save(f'model_{name}', **arrs)
        print(name, 'done in %.1fs' % (time.time() - t0))
        del net
    with open(os.path.join(HERE, 'schema.json'), 'w') as f:
        json.dump(schemas, f)


def gen_kdiff():
    """A training step whose contrastive loss is O(1): with name-seeded weights the encoder's embedding is all but
    input-independent, so with encoder_k == encoder_q the InfoNCE term is 6e-6 whatever the key image is and the encoder's gradients are
    ~1e-7.  Here the key encoder keeps its OWN name-seeded weights (no q -> k copy): l_pos is the cosine of two unrelated embeddings,
    the contrastive term is ~1.3 and the query encoder's backward is pinned at real magnitudes.  all_3_bands / L = 3 / freq, batch 2."""
    set_opt(batch_size=2, degradation_embedding_method=['all_3_bands'], L=3, encoder_msa_type='freq')
    net = seed_module(AirNet(opt), '')
    clean, q, k = synth_batch(2, 128, 'model.')
    net.train()
    restored, logits, labels = net(x_query=q, x_key=k)
    CE = torch.nn.CrossEntropyLoss()
    contrast = sum(CE(logits[i], labels[i]) for i in range(opt.L)) / opt.L
    l1 = torch.nn.L1Loss()(restored, clean)
    loss = l1 + opt.contrast_loss_weight * contrast
    g = grads_of(net, loss)
    names = sorted(g.keys())
    arrs = {'restored_train': restored, 'logits': torch.stack(logits, 0), 'loss': loss, 'l1': l1, 'contrast': contrast,
            'grad_names': np.array(names), 'grad_norms': np.array([g[n].norm().item() for n in names]),
            'queue_after': net.E.E.queue, 'queue_ptr_after': net.E.E.queue_ptr}
    for n in names:
        if n.startswith('E.E.encoder_q.') and g[n].numel() <= 4096 and ('blocks.0.' in n or 'blocks.1.' in n or 'mlp.' in n or 'norm' in n):
            arrs['g.' + n] = g[n]
    for n in ('E.E.encoder_q.uformer.input_proj.proj.0.weight', 'E.E.encoder_q.uformer.dowsample_0.conv.0.weight',
              'E.E.encoder_q.uformer.encoderlayer_1.blocks.1.attn_inter.qkv.to_kv.weight', 'R.R.output_proj.proj.0.weight'):
        arrs['g.' + n] = g[n]
    save('model_all3_kdiff', **arrs)
    print('kdiff: contrast %.4f' % float(contrast))


def gen_model256():
    """Resolution-generic row: the reference's own classes built with img_size=256 -- the seam `cls(opt)` always
    takes the 128 default (net/model.py:17,31), so the two registered names are rebound to 256-pixel constructors for this run.
    One image, all_3_bands / L=3 / freq: eval output, train-mode restored / logits / loss, gradient norms."""
    import net.model as RM
    set_opt(batch_size=1, degradation_embedding_method=['all_3_bands'], L=3, encoder_msa_type='freq')
    keep = RM.UformerEncoder, RM.UformerDecoder
    RM.UformerEncoder = lambda o: RE.UformerEncoder(o, img_size=256)
    RM.UformerDecoder = lambda o: RD.UformerDecoder(o, img_size=256)
    try:
        t0 = time.time()
        net = seed_module(AirNet(opt), '')
    finally:
        RM.UformerEncoder, RM.UformerDecoder = keep
    for pq, pk in zip(net.E.E.encoder_q.parameters(), net.E.E.encoder_k.parameters()):
        pk.data.copy_(pq.data)
    clean, q, k = synth_batch(1, 256, 'model256.')
    net.eval()
    with torch.no_grad():
        restored_eval = net(x_query=q, x_key=q)
    arrs = {'restored_eval': restored_eval, 'psnr_eval': O.psnr(restored_eval, clean), 'psnr_input': O.psnr(q, clean)}
    net.train()
    restored, logits, labels = net(x_query=q, x_key=k)
    CE = torch.nn.CrossEntropyLoss()
    contrast = sum(CE(logits[i], labels[i]) for i in range(opt.L)) / opt.L
    l1 = torch.nn.L1Loss()(restored, clean)
    loss = l1 + opt.contrast_loss_weight * contrast
    g = grads_of(net, loss)
    names = sorted(g.keys())
    arrs.update({'restored_train': restored, 'logits': torch.stack(logits, 0), 'loss': loss, 'l1': l1, 'contrast': contrast,
                 'grad_names': np.array(names), 'grad_norms': np.array([g[n].norm().item() for n in names]),
                 'queue_after': net.E.E.queue, 'queue_ptr_after': net.E.E.queue_ptr})
    for n in ('R.R.output_proj.proj.0.weight', 'R.R.input_proj.proj.0.weight', 'E.E.encoder_q.uformer.input_proj.proj.0.weight',
              'R.R.bottleneck_0.blocks.1.attn.relative_position_bias_table'):
        arrs['g.' + n] = g[n]
    save('model256_all3', **arrs)
    print('model256 done in %.1fs' % (time.time() - t0))


def gen_debug():
    """`opt.debug_mode = True` (plot_MSA_frequency.py:47, plot_embed_lamb_curve.py:48): the decoder returns (restored, visual_freqs),
    visual_freqs[layer][block] = [spectrum of LN(x) before the attention, spectrum of the attention output, embed_lamb]
    (decoder_Uformer.py:668-673,731-736,753-756,1168-1169).  Eval mode, all_3_bands, the seeded model of `model_all3`."""
    set_opt(batch_size=2, degradation_embedding_method=['all_3_bands'], L=3, encoder_msa_type='freq', debug_mode=True)
    try:
        net = seed_module(AirNet(opt), '')
    finally:
        set_opt(debug_mode=False)
    clean, q, k = synth_batch(2, 128, 'model.')
    net.eval()
    with torch.no_grad():
        fea, inter = net.E(x_query=q, x_key=q)
        restored, vf = net.R(x_query=q, inter=inter)
    arrs = {'restored': restored, 'layers': np.array([len(layer) for layer in vf])}
    for li, layer in enumerate(vf):
        for bi, (before, after, lamb) in enumerate(layer):
            arrs[f'before.{li}.{bi}'] = before
            arrs[f'after.{li}.{bi}'] = after
            arrs[f'lamb.{li}.{bi}'] = lamb
    save('debug_all3', **arrs)


def gen_convnets():
    """The convolutional plug-ins of the seam (BASELINE configs[0]): everything of net/encoder_ResNet.py and net/decoder_DGRN.py that
    RUNS in the reference -- ResBlock (stride 1 and 2, train mode: batch statistics), ResNetEncoder (eval and train, standalone),
    SFT_layer -- forward + backward.  DCN_layer.forward ends in `assert False` (deform_conv.py:64; mmcv absent), so DGM / DGB / DGG /
    DGRN.forward cannot be executed and are NOT in any golden: their wiring is checked against the oracle's restatement only and
    the deformable convolution by known-answer tests (parity unpinned)."""
    from net import encoder_ResNet as RR
    from net import decoder_DGRN as RG
    arrs = {}
    for tag, cin, cout, stride, hw in (('s1', 3, 64, 1, 32), ('s2', 64, 128, 2, 16)):
        pre = f'unit_resblock_{tag}.'
        blk = seed_module(RR.ResBlock(cin, cout, stride), pre).train()
        x = rnd(pre + 'x', (2, cin, hw, hw)).requires_grad_(True)
        y = blk(x)
        w = rnd(pre + 'dy', y.shape)
        g = grads_of(blk, (y * w).sum())
        a = {'x': x, 'y': y, 'dy': w, 'dx': x.grad}
        a.update({'g.' + k: v for k, v in g.items()})
        a.update({'s.' + k: v for k, v in blk.state_dict().items() if 'running' in k})
        save(f'unit_resblock_{tag}', **a)
    pre = 'unit_sft.'
    sft = seed_module(RG.SFT_layer(64, 64), pre)
    x = rnd(pre + 'x', (2, 64, 16, 16)).requires_grad_(True)
    it = rnd(pre + 'inter', (2, 64, 16, 16)).requires_grad_(True)
    y = sft(x, it)
    w = rnd(pre + 'dy', y.shape)
    g = grads_of(sft, (y * w).sum())
    a = {'x': x, 'inter': it, 'y': y, 'dy': w, 'dx': x.grad, 'dinter': it.grad}
    a.update({'g.' + k: v for k, v in g.items()})
    save('unit_sft', **a)
    # the whole encoder, standalone (the MoCo wrapper indexes L heads on its 1-head output in train mode, moco.py:127-128)
    set_opt(encoder_type='ResNet', decoder_type='ResNet', encoder_dim=256, batch_size=2)
    try:
        pre = 'E.E.encoder_q.'
        enc = seed_module(RR.ResNetEncoder(opt), pre)
        x = rnd('resnet.x', (2, 3, 32, 32), 0.5)
        enc.eval()
        with torch.no_grad():
            fea, out, inter = enc(x)
        a = {'x': x, 'fea_eval': fea, 'out_eval': out[0], 'inter_eval': inter}
        enc.train()
        fea, out, inter = enc(x)
        w1, w2 = rnd('resnet.dout', out[0].shape), rnd('resnet.dinter', inter.shape)
        g = grads_of(enc, (out[0] * w1).sum() + (inter * w2).sum())
        names = sorted(g.keys())
        a.update({'fea_train': fea, 'out_train': out[0], 'inter_train': inter, 'dout': w1, 'dinter': w2,
                  'grad_names': np.array(names), 'grad_norms': np.array([g[n].norm().item() for n in names])})
        a.update({'g.' + k: v for k, v in g.items() if v.numel() <= 40000})
        a.update({'s.' + k: v for k, v in enc.state_dict().items() if 'running' in k})
        save('model_resnet_encoder', **a)
        net = AirNet(opt)
        sch = json.load(open(os.path.join(HERE, 'schema.json')))
        sch['resnet_dgrn'] = schema_of(net)
        json.dump(sch, open(os.path.join(HERE, 'schema.json'), 'w'))
    finally:
        set_opt(encoder_type='Uformer', decoder_type='Uformer')


def gen_vit():
    """ViT encoder (BASELINE configs[4]; encoder_ViT.py:17-203) standalone at 128x128: eval (fea, out, inter), and train mode with every
    Dropout set to p = 0 (stochastic otherwise): outputs + all parameter gradients.  Plus the one end-to-end configuration with it
    that runs in the reference: ViT encoder + plain Uformer decoder, eval forward."""
    from net import encoder_ViT as RV
    set_opt(encoder_type='ViT', decoder_type='Uformer', encoder_dim=3, batch_size=2, degradation_embedding_method=['None'],
            frequency_decompose_type='none', out_channels=3, batch_wise_decompose=False)
    try:
        pre = 'E.E.encoder_q.'
        enc = seed_module(RV.ViTEncoder(opt), pre)
        for m in enc.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        x = rnd('vit.x', (2, 3, 128, 128), 0.5)
        enc.eval()
        with torch.no_grad():
            fea, out, inter = enc(x)
        a = {'x': x, 'fea_eval': fea, 'out_eval': out[0], 'inter_eval': inter}
        enc.train()
        fea, out, inter = enc(x)
        w1, w2 = rnd('vit.dout', out[0].shape), rnd('vit.dinter', inter.shape)
        g = grads_of(enc, (out[0] * w1).sum() + (inter * w2).sum())
        names = sorted(g.keys())
        a.update({'fea_train': fea, 'out_train': out[0], 'inter_train': inter, 'dout': w1, 'dinter': w2,
                  'grad_names': np.array(names), 'grad_norms': np.array([g[n].norm().item() for n in names])})
        for n in names:
            if g[n].numel() <= 4096 or n in ('pos_embedding',):
                a['g.' + n] = g[n]
        a.update({'s.' + k: v for k, v in enc.state_dict().items() if 'running' in k})
        save('model_vit_encoder', **a)
        net = seed_module(AirNet(opt), '')
        for pq, pk in zip(net.E.E.encoder_q.parameters(), net.E.E.encoder_k.parameters()):
            pk.data.copy_(pq.data)
        sch = json.load(open(os.path.join(HERE, 'schema.json')))
        sch['vit_uformer'] = schema_of(net)
        json.dump(sch, open(os.path.join(HERE, 'schema.json'), 'w'))
        clean, q, k = synth_batch(2, 128, 'model.')
        net.eval()
        with torch.no_grad():
            restored = net(x_query=q, x_key=q)
        save('model_vit_uformer', restored_eval=restored, psnr_eval=O.psnr(restored, clean))
    finally:
        set_opt(encoder_type='Uformer', decoder_type='Uformer', encoder_dim=256, degradation_embedding_method=['all_3_bands'])


DROP_SEED = 20240917


def patch_dropout(seed):
    """torch.nn.Dropout.forward of the IMPORTED reference draws the product's counter-based masks (oracle/dropout_hash.py) at the
    call sites numbered by `assign_vit_sites`: train-mode goldens with Dropout ON become deterministic statements.  torch's own
    Philox stream cannot be matched by a HIP kernel (it differs between torch's CPU and GPU generators too); the Bernoulli(1-p)/(1-p)
    arithmetic is untouched."""
    import dropout_hash as DH

    def forward(self, x):
        site = getattr(self, '_site', None)
        if not self.training or self.p == 0 or site is None:
            return x
        m = torch.from_numpy(DH.keep_mask(seed, site, tuple(x.shape), self.p)).to(x.dtype)
        return x * m / (1.0 - self.p)
    torch.nn.Dropout.forward = forward


def assign_vit_sites(enc, prefix):
    import dropout_hash as DH
    base = DH.site_base(prefix)
    enc.dropout._site = DH.vit_site(base, 0, 'emb')
    for i, (attn, ff) in enumerate(enc.transformer.layers):
        attn.fn.dropout._site = DH.vit_site(base, i, 'attn')
        attn.fn.to_out[1]._site = DH.vit_site(base, i, 'out')
        ff.fn.net[2]._site = DH.vit_site(base, i, 'hidden')
        ff.fn.net[4]._site = DH.vit_site(base, i, 'ff')


def vit_encoder_arrays(enc, x, tag, sub):
    """eval outputs + train-mode (Dropout ON, hashed masks) outputs and gradients of a reference ViTEncoder; `inter` maps are kept
    at every sub-th pixel (fixture size)."""
    enc.eval()
    with torch.no_grad():
        fea, out, inter = enc(x)
    a = {'fea_eval': fea, 'out_eval': out[0], 'inter_eval': inter[:, :, ::sub, ::sub]}         # x is seeded: rnd(tag + 'x')
    enc.train()
    fea, out, inter = enc(x)
    w1, w2 = rnd(tag + 'dout', out[0].shape), rnd(tag + 'dinter', inter.shape)
    g = grads_of(enc, (out[0] * w1).sum() + (inter * w2).sum())
    names = sorted(g.keys())
    a.update({'fea_train': fea, 'out_train': out[0], 'inter_train': inter[:, :, ::sub, ::sub],
              'grad_names': np.array(names), 'grad_norms': np.array([g[n].norm().item() for n in names])})
    for n in names:
        if g[n].numel() <= 4096:
            a['g.' + n] = g[n]
    a.update({'s.' + k: v for k, v in enc.state_dict().items() if 'running' in k})
    return a


def gen_vit256():
    """BASELINE configs[4]: the reference's ViTEncoder constructed with image_size=256 (N = 256 tokens, pos_embedding [1, 256, 768]):
    eval, and TRAIN mode with every Dropout at its p = 0.1 (hashed masks, see patch_dropout); plus ViT(256) + plain Uformer decoder
    built with img_size=256, eval forward (the seam's names are rebound to the 256-pixel constructors, as gen_model256 does)."""
    from net import encoder_ViT as RV
    import net.model as RM
    set_opt(encoder_type='ViT', decoder_type='Uformer', encoder_dim=3, batch_size=2, degradation_embedding_method=['None'],
            frequency_decompose_type='none', out_channels=3, batch_wise_decompose=False)
    keep_fwd = torch.nn.Dropout.forward
    try:
        patch_dropout(DROP_SEED)
        pre = 'E.E.encoder_q.'
        enc = seed_module(RV.ViTEncoder(opt, image_size=256), pre)
        assign_vit_sites(enc, pre)
        x = rnd('vit256.x', (2, 3, 256, 256), 0.5)
        save('model_vit256_encoder', drop_seed=np.int64(DROP_SEED), **vit_encoder_arrays(enc, x, 'vit256.', 4))
        keep = RM.ViTEncoder, RM.UformerDecoder
        RM.ViTEncoder = lambda o: RV.ViTEncoder(o, image_size=256)
        RM.UformerDecoder = lambda o: RD.UformerDecoder(o, img_size=256)
        try:
            set_opt(batch_size=1)
            net = seed_module(AirNet(opt), '')
        finally:
            RM.ViTEncoder, RM.UformerDecoder = keep
        for pq, pk in zip(net.E.E.encoder_q.parameters(), net.E.E.encoder_k.parameters()):
            pk.data.copy_(pq.data)
        sch = json.load(open(os.path.join(HERE, 'schema.json')))
        sch['vit256_uformer'] = schema_of(net)
        json.dump(sch, open(os.path.join(HERE, 'schema.json'), 'w'))
        clean, q, k = synth_batch(1, 256, 'model256.')
        net.eval()
        with torch.no_grad():
            restored = net(x_query=q, x_key=q)
        save('model_vit256_uformer', restored_eval=restored[:, :, ::2, ::2], psnr_eval=O.psnr(restored, clean))
    finally:
        torch.nn.Dropout.forward = keep_fwd
        set_opt(encoder_type='Uformer', decoder_type='Uformer', encoder_dim=256, degradation_embedding_method=['all_3_bands'], batch_size=2)


def gen_vit_lamb():
    """The learnable band re-weighting of the ViT attention maps (encoder_ViT.py:51-66,85-92) at 128x128 (N = dim_head = 64, where the
    reference runs it): frequency_decompose_type '3_bands', 'DC' and batch-wise 'DC'; eval + train (Dropout ON, hashed masks);
    `lamb` is seeded non-zero like every other parameter (the reference initialises it to zeros)."""
    from net import encoder_ViT as RV
    keep_fwd = torch.nn.Dropout.forward
    try:
        patch_dropout(DROP_SEED)
        for tag, ftype, bw in (('3bands', '3_bands', False), ('DC', 'DC', False), ('DCbw', 'DC', True)):
            set_opt(encoder_type='ViT', decoder_type='Uformer', encoder_dim=3, batch_size=2, degradation_embedding_method=['None'],
                    frequency_decompose_type=ftype, out_channels=3, batch_wise_decompose=bw)
            pre = 'E.E.encoder_q.'
            enc = seed_module(RV.ViTEncoder(opt), pre)
            assign_vit_sites(enc, pre)
            x = rnd('vitlamb.x', (2, 3, 128, 128), 0.5)
            save(f'model_vit_lamb_{tag}', drop_seed=np.int64(DROP_SEED), **vit_encoder_arrays(enc, x, 'vitlamb.', 4))
    finally:
        torch.nn.Dropout.forward = keep_fwd
        set_opt(encoder_type='Uformer', decoder_type='Uformer', encoder_dim=256, degradation_embedding_method=['all_3_bands'], batch_size=2,
                frequency_decompose_type='none', batch_wise_decompose=False)


def gen_moco():
    """Three consecutive train-mode steps of the encoder side only (net.E), SGD lr 0.05 on the
    query encoder in between, so that EMA, queue rotation and pointer wrap are all exercised."""
    set_opt(batch_size=2, degradation_embedding_method=['all_3_bands'], L=3, encoder_msa_type='freq')
    net = seed_module(AirNet(opt), '')
    for pq, pk in zip(net.E.E.encoder_q.parameters(), net.E.E.encoder_k.parameters()):
        pk.data.copy_(pq.data)
    net.train()
    arrs = {}
    CE = torch.nn.CrossEntropyLoss()
    probe = ['uformer.input_proj.proj.0.weight', 'mlp.0.2.weight', 'uformer.conv.blocks.1.mlp.linear2.0.bias']
    for step in range(4):
        _, q, k = synth_batch(2, 128, f'moco{step}.')
        _, logits, labels, inter = net.E(x_query=q, x_key=k)
        loss = sum(CE(logits[i], labels[i]) for i in range(3)) / 3
        for p in net.parameters():
            p.grad = None
        loss.backward()
        with torch.no_grad():
            for p in net.E.E.encoder_q.parameters():
                if p.grad is not None:
                    p -= 0.05 * p.grad
        arrs[f'logits{step}'] = torch.stack(logits, 0)
        arrs[f'loss{step}'] = loss
        arrs[f'queue{step}'] = net.E.E.queue.clone()
        arrs[f'ptr{step}'] = net.E.E.queue_ptr.clone()
        ksd = net.E.E.encoder_k.state_dict()
        for n in probe:
            arrs[f'k{step}.' + n] = ksd[n].clone()
    save('moco_steps', **arrs)


if __name__ == '__main__':
    what = _ARGV or ['unit', 'model', 'moco']
    with torch.enable_grad():
        if 'unit' in what:
            gen_unit()
        if 'model' in what:
            gen_model()
        if 'moco' in what:
            gen_moco()
        if 'model256' in what:
            gen_model256()
        if 'debug' in what:
            gen_debug()
        if 'convnets' in what:
            gen_convnets()
        if 'vit' in what:
            gen_vit()
        if 'vit256' in what:
            gen_vit256()
        if 'kdiff' in what:
            gen_kdiff()
        if 'vitlamb' in what:
            gen_vit_lamb()
