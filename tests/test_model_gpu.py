"""GPU parity of the whole MI355X-native AirNet (HIP kernels behind net.model.AirNet) against
  (a) goldens produced by the REAL reference (tests/golden/model_*.npz), and
  (b) the CPU oracle on the same seeded weights / inputs.
fp32 compute: tensors within 1e-4 relative (north_star), PSNR within 0.01 dB.  bf16 compute: PSNR within
0.01 dB of the reference value, loss within 2 %.  DropPath is neutralised exactly as in the golden run."""
import pytest
import torch

import airnet_oracle as O
from helpers import VARIANTS, close, load, make_opt, schema, synth_batch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def build(variant, dtype='fp32', batch_size=2):
    from net.model import AirNet
    from fwair import functional as Fn
    opt = make_opt(variant, batch_size=batch_size, compute_dtype=dtype)
    net = AirNet(opt)
    st = O.fill_state_seeded(schema(variant))
    sd = net.state_dict()
    for k in sd:
        if st.get(k) is not None and sd[k].is_floating_point():
            sd[k] = st[k]
    net.load_state_dict(sd)
    Fn.set_droppath_override(lambda name, n, rate, device: None)      # DropPath off (goldens were made that way)
    return net.to(DEV), opt


@pytest.mark.parametrize('variant', list(VARIANTS))
def test_eval_forward_fp32(variant):
    g = load(f'model_{variant}')
    net, opt = build(variant)
    clean, q, k = synth_batch(2, 128, 'model.')
    net.eval()
    with torch.no_grad():
        out = net(x_query=q.to(DEV), x_key=q.to(DEV))
    err = close(out, g['restored_eval'], 1e-4, 'restored_eval vs reference golden')
    assert abs(O.psnr(out.cpu(), clean) - float(g['psnr_eval'])) < 0.01
    print(f'{variant}: eval rel err {err:.2e}')


@pytest.mark.parametrize('variant', ['all3', 'all2_L2'])
def test_train_step_fp32(variant):
    g = load(f'model_{variant}')
    net, opt = build(variant)
    clean, q, k = synth_batch(2, 128, 'model.')
    net.train()
    restored, logits, labels = net(x_query=q.to(DEV), x_key=k.to(DEV))
    close(restored, g['restored_train'], 1e-4, 'restored_train')
    close(torch.stack(logits), g['logits'], 2e-4, 'logits')
    CE = torch.nn.CrossEntropyLoss()
    contrast = sum(CE(logits[i], labels[i]) for i in range(opt.L)) / opt.L
    loss = torch.nn.L1Loss()(restored, clean.to(DEV)) + opt.contrast_loss_weight * contrast
    close(loss, g['loss'], 1e-4, 'loss')
    loss.backward()
    names = [str(n) for n in g['grad_names']]
    params = dict(net.named_parameters())
    norms = torch.tensor([params[n].grad.norm().item() for n in names])
    # relative deviation, with a floor of 1e-6 x the largest norm: gradients that are ~0 by cancellation
    # (e.g. 5e-11 for a 1x1 lambda-MLP weight) carry no relative information
    floor = float(g['grad_norms'].max()) * 1e-6
    rel = ((norms - g['grad_norms']).abs() / g['grad_norms'].clamp_min(floor))
    worst = int(rel.argmax())
    assert rel.max() < 5e-3, f'grad norm of {names[worst]}: {norms[worst]:.6e} vs {g["grad_norms"][worst]:.6e}'
    gmax = float(g['grad_norms'].max())
    for key, val in g.items():
        if key.startswith('g.'):     # tensors that are ~0 by cancellation (1e-9 lambda-MLP grads) only see float-atomic ordering noise
            close(params[key[2:]].grad, val, 5e-3 if float(val.norm()) > 1e-6 * gmax else 5e-2, key)
    close(net.E.E.queue, g['queue_after'], 1e-4, 'queue')
    assert int(net.E.E.queue_ptr) == int(g['queue_ptr_after'])
    close(net.E.E.encoder_q.norm[0][0].running_mean, g['bn_q0_running_mean'], 1e-3, 'bn running mean (q)')
    close(net.E.E.encoder_q.norm[0][0].running_var, g['bn_q0_running_var'], 1e-3, 'bn running var (q)')
    close(net.E.E.encoder_k.norm[0][0].running_mean, g['bn_k0_running_mean'], 1e-3, 'bn running mean (k)')


def test_eval_and_train_bf16():
    g = load('model_all3')
    net, opt = build('all3', 'bf16')
    clean, q, k = synth_batch(2, 128, 'model.')
    net.eval()
    with torch.no_grad():
        out = net(x_query=q.to(DEV), x_key=q.to(DEV))
    assert torch.isfinite(out).all()
    assert abs(O.psnr(out.cpu(), clean) - float(g['psnr_eval'])) < 0.01, (O.psnr(out.cpu(), clean), float(g['psnr_eval']))
    close(out, g['restored_eval'], 2e-2, 'bf16 restored_eval')
    net.train()
    restored, logits, labels = net(x_query=q.to(DEV), x_key=k.to(DEV))
    CE = torch.nn.CrossEntropyLoss()
    contrast = sum(CE(logits[i], labels[i]) for i in range(opt.L)) / opt.L
    loss = torch.nn.L1Loss()(restored, clean.to(DEV)) + opt.contrast_loss_weight * contrast
    assert abs(float(loss) - float(g['loss'])) / float(g['loss']) < 2e-2
    loss.backward()
    names = [str(n) for n in g['grad_names']]
    params = dict(net.named_parameters())
    norms = torch.tensor([params[n].grad.norm().item() for n in names])
    assert torch.isfinite(norms).all()
    rel = ((norms - g['grad_norms']).abs() / g['grad_norms'].clamp_min(1e-12))
    big = g['grad_norms'] > g['grad_norms'].max() * 1e-3
    assert rel[big].median() < 5e-2, f'median relative grad-norm deviation {rel[big].median():.3e}'


def test_moco_three_steps_fp32():
    """EMA, queue rotation and pointer wrap over 4 encoder-only steps with SGD in between (tests/golden/moco_steps.npz)."""
    g = load('moco_steps')
    net, opt = build('all3')
    net.train()
    CE = torch.nn.CrossEntropyLoss()
    probe = ['uformer.input_proj.proj.0.weight', 'mlp.0.2.weight', 'uformer.conv.blocks.1.mlp.linear2.0.bias']
    for step in range(4):
        _, q, k = synth_batch(2, 128, f'moco{step}.')
        _, logits, labels, inter = net.E(x_query=q.to(DEV), x_key=k.to(DEV))
        loss = sum(CE(logits[i], labels[i]) for i in range(3)) / 3
        for p in net.parameters():
            p.grad = None
        loss.backward()
        with torch.no_grad():
            for p in net.E.E.encoder_q.parameters():
                if p.grad is not None:
                    p -= 0.05 * p.grad
        close(torch.stack(logits), g[f'logits{step}'], 1e-3, f'logits step {step}')
        close(net.E.E.queue, g[f'queue{step}'], 1e-3, f'queue step {step}')
        assert int(net.E.E.queue_ptr) == int(g[f'ptr{step}'])
        ksd = net.E.E.encoder_k.state_dict()
        for n in probe:
            close(ksd[n], g[f'k{step}.' + n], 1e-4, f'EMA {n} step {step}')


def test_engine_two_stage_backward_matches_single_pass():
    """Data-parallel replay scheme (decoder backward | all-reduce of its gradients overlapping the encoder backward | optimizer)
    forced on one GPU: same losses and same parameters as the single-graph step after 2 steps, fp32, DropPath off."""
    from fwair import engine as E
    from fwair import functional as Fn
    res = []
    for split in (False, True):
        net, opt = build('all3')
        net.train()
        eng = E.TrainEngine(net, lr=2e-4, contrast_loss_weight=0.6, use_graph=True, split_backward=split)
        clean, q, k = synth_batch(2, 128, 'split.')
        outs = [eng.step(q.to(DEV), k.to(DEV), clean.to(DEV)).clone() for _ in range(2)]
        torch.cuda.synchronize()
        assert (eng._gsplit is not None) == split
        res.append((torch.stack(outs).cpu(), eng.flat_p.clone().cpu(), eng.flat_g.clone().cpu()))
        Fn.config.direct_grads = False
    (la, pa, ga), (lb, pb, gb) = res
    close(lb, la, 1e-5, 'losses (total, l1, contrast) over 2 steps')
    close(gb, ga, 2e-4, 'flat gradient of the second step')
    # Adam normalises every gradient to about +-lr, so where a gradient is numerically zero the order of the float atomics decides
    # its sign: compare the bulk, not the maximum
    assert float((pb - pa).abs().median()) < 1e-7 and float(((pb - pa).abs() > 1e-5).float().mean()) < 0.02, 'parameters after 2 Adam steps'


def test_eval_after_graph_replayed_steps_sees_current_weights():
    """A captured step changes the weights without running any Python, so the on-demand re-laid-out weight copies of
    functional.shadow() (depthwise taps, k4 / transposed conv weights, narrow rows) must not be trusted across a replay: an
    eval forward right after training equals the same forward with every cached copy invalidated by hand."""
    from fwair import engine as E
    from fwair import functional as Fn
    net, opt = build('all3', 'bf16')
    net.train()
    eng = E.TrainEngine(net, lr=1e-2, contrast_loss_weight=0.6, use_graph=True)      # large steps: stale copies would show
    clean, q, k = synth_batch(2, 128, 'stale.')
    for _ in range(3):
        eng.step(q.to(DEV), k.to(DEV), clean.to(DEV))
    net.eval()
    with torch.no_grad():
        a = net(x_query=q.to(DEV), x_key=q.to(DEV)).clone()
        Fn.config.shadow_epoch += 1
        b = net(x_query=q.to(DEV), x_key=q.to(DEV))
    Fn.config.direct_grads = False
    assert torch.equal(a, b)


def test_graph_steps_survive_eval_and_empty_cache():
    """ADVICE r2 (use-after-free of weight shadows behind captured graphs): graph steps, an eval forward (which re-derives the
    re-laid-out weight copies -- it must do so INTO the buffers the graph captured), torch.cuda.empty_cache(), a foreign
    allocation, then more graph steps: losses and parameters equal an eager engine fed the same sequence."""
    from fwair import engine as E
    from fwair import functional as Fn
    clean, q, k = (t.to(DEV) for t in synth_batch(2, 128, 'uaf.'))

    def run(graph):
        net, opt = build('all3', 'bf16')
        net.train()
        eng = E.TrainEngine(net, lr=1e-3, contrast_loss_weight=0.6, use_graph=graph)
        losses = []
        for _ in range(2):
            losses.append(eng.step(q, k, clean).clone())
        net.eval()
        with torch.no_grad():
            ev = net(x_query=q, x_key=q).clone()
        net.train()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        junk = [torch.full((1 << 22,), float('nan'), device=DEV) for _ in range(16)]      # reuse whatever was freed
        for _ in range(2):
            losses.append(eng.step(q, k, clean).clone())
        torch.cuda.synchronize()
        del junk
        flat = eng.flat_p.clone()
        Fn.config.direct_grads = False
        return torch.stack(losses), ev, flat

    la, eva, pa = run(True)
    lb, evb, pb = run(False)
    assert torch.isfinite(la).all() and torch.isfinite(pa).all()
    # bf16 + Adam: a gradient that is ~0 may change sign between two runs (float atomics), and Adam turns either sign into a full
    # +-lr step -- two correct runs differ by up to 2 lr per parameter and step; a stale / freed operand copy gives errors of O(1) or NaN
    close(la, lb, 5e-3, 'losses graph vs eager across an eval + empty_cache')
    close(eva, evb, 2e-2, 'eval output between the steps')
    assert float((pa - pb).abs().max()) <= 4 * 2 * 1e-3 + 1e-6, 'parameters after the fourth step'


def test_256_resolution_fp32_and_bf16():
    """Resolution-generic construction: `opt.patch_size=256` through the unchanged seam, against the golden made by
    the reference's classes with img_size=256 (16x16 bottleneck with shifted odd blocks, 256-token LFS heads, 256-point band DFT)."""
    from net.model import AirNet
    from fwair import functional as Fn
    g = load('model256_all3')
    clean, q, k = synth_batch(1, 256, 'model256.')
    st = O.fill_state_seeded(schema('all3'))
    st['E.E.queue'] = torch.nn.functional.normalize(O.seeded_tensor('E.E.queue', (3, 256, 3)) / 0.02, dim=1)   # K = 3 * batch_size

    def make(dtype):
        opt = make_opt('all3', batch_size=1, patch_size=256, compute_dtype=dtype)
        net = AirNet(opt)
        sd = net.state_dict()
        for key in sd:
            if st.get(key) is not None and sd[key].is_floating_point():
                sd[key] = st[key]
        net.load_state_dict(sd)
        Fn.set_droppath_override(lambda name, n, rate, device: None)
        return net.to(DEV), opt

    net, opt = make('fp32')
    net.eval()
    with torch.no_grad():
        out = net(x_query=q.to(DEV), x_key=q.to(DEV))
    close(out, g['restored_eval'], 1e-4, 'restored_eval (256) vs reference golden')
    assert abs(O.psnr(out.cpu(), clean) - float(g['psnr_eval'])) < 0.01
    net.train()
    restored, logits, labels = net(x_query=q.to(DEV), x_key=k.to(DEV))
    close(restored, g['restored_train'], 1e-4, 'restored_train (256)')
    close(torch.stack(logits), g['logits'], 2e-4, 'logits (256)')
    CE = torch.nn.CrossEntropyLoss()
    contrast = sum(CE(logits[i], labels[i]) for i in range(opt.L)) / opt.L
    loss = torch.nn.L1Loss()(restored, clean.to(DEV)) + opt.contrast_loss_weight * contrast
    close(loss, g['loss'], 1e-4, 'loss (256)')
    loss.backward()
    names = [str(n) for n in g['grad_names']]
    params = dict(net.named_parameters())
    norms = torch.tensor([params[n].grad.norm().item() for n in names])
    # floor of 1e-5 x the largest norm (0.77): the lambda-head gradients (1e-12 ... 6e-7 here) all hang off one scalar per
    # (block, band, head) that sums 64x64 cancelling terms over 4x as many windows as at 128x128; they move by ~1 % with the
    # order of the float atomics (and the reference's own CPU summation order) while every other gradient agrees to < 1e-3
    floor = float(g['grad_norms'].max()) * 1e-5
    rel = ((norms - g['grad_norms']).abs() / g['grad_norms'].clamp_min(floor))
    worst = int(rel.argmax())
    assert rel.max() < 5e-3, f'grad norm of {names[worst]}: {norms[worst]:.6e} vs {g["grad_norms"][worst]:.6e}'
    gmax = float(g['grad_norms'].max())
    for key, val in g.items():
        if key.startswith('g.'):
            close(params[key[2:]].grad, val, 5e-3 if float(val.norm()) > 1e-6 * gmax else 5e-2, key)
    close(net.E.E.queue, g['queue_after'], 1e-4, 'queue (256)')
    del net, params
    net, opt = make('bf16')
    net.eval()
    with torch.no_grad():
        out = net(x_query=q.to(DEV), x_key=q.to(DEV))
    assert abs(O.psnr(out.float().cpu(), clean) - float(g['psnr_eval'])) < 0.01
    net.train()
    restored, logits, labels = net(x_query=q.to(DEV), x_key=k.to(DEV))
    contrast = sum(CE(logits[i], labels[i]) for i in range(opt.L)) / opt.L
    loss = torch.nn.L1Loss()(restored, clean.to(DEV)) + opt.contrast_loss_weight * contrast
    assert abs(float(loss) - float(g['loss'])) / float(g['loss']) < 2e-2
    loss.backward()
    params = dict(net.named_parameters())
    norms = torch.tensor([params[n].grad.norm().item() for n in names])
    assert torch.isfinite(norms).all()
    rel = ((norms - g['grad_norms']).abs() / g['grad_norms'].clamp_min(1e-12))
    big = g['grad_norms'] > g['grad_norms'].max() * 1e-3
    assert rel[big].median() < 5e-2, f'median relative grad-norm deviation {rel[big].median():.3e}'


def test_debug_mode_payload():
    """`opt.debug_mode = True` (plot_MSA_frequency.py:47, plot_embed_lamb_curve.py:48) -> the decoder returns
    (restored, visual_freqs), visual_freqs[layer][block] = [spectrum_before [H, W], spectrum_after [H, W], embed_lamb [B, 1, heads]]
    (decoder_Uformer.py:668-673,731-736,753-756,1168-1169); against the golden produced by the reference in debug mode."""
    from net.model import AirNet
    from fwair import functional as Fn
    g = load('debug_all3')
    opt = make_opt('all3', debug_mode=True)
    net = AirNet(opt)
    st = O.fill_state_seeded(schema('all3'))
    sd = net.state_dict()
    assert len(sd) == 2496                                    # the debug modules add no state
    for k in sd:
        if st.get(k) is not None and sd[k].is_floating_point():
            sd[k] = st[k]
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    clean, q, k = synth_batch(2, 128, 'model.')
    with torch.no_grad():
        fea, inter = net.E(x_query=q.to(DEV), x_key=q.to(DEV))
        restored, vf = net.R(x_query=q.to(DEV), inter=inter)
        both = net(x_query=q.to(DEV), x_key=q.to(DEV))        # AirNet.forward hands the tuple through (net/model.py:66-71)
    assert isinstance(both, tuple) and len(both) == 2
    close(restored, g['restored'], 1e-4, 'restored (debug mode)')
    assert [len(layer) for layer in vf] == g['layers'].tolist()
    worst = 0.0
    for li, layer in enumerate(vf):
        for bi, (before, after, lamb) in enumerate(layer):
            assert before.shape == g[f'before.{li}.{bi}'].shape
            worst = max(worst, close(before, g[f'before.{li}.{bi}'], 1e-4, f'spectrum before {li}.{bi}'))
            worst = max(worst, close(after, g[f'after.{li}.{bi}'], 1e-4, f'spectrum after {li}.{bi}'))
            worst = max(worst, close(lamb, g[f'lamb.{li}.{bi}'], 1e-4, f'embed_lamb {li}.{bi}'))
    print(f'debug payload: worst rel-to-max err {worst:.2e}')
    # the plain decoder (degradation_embedding_method None, as plot_MSA_frequency.py:42 sets it): embed_lamb is [] there
    opt2 = make_opt('all3', debug_mode=True, degradation_embedding_method=['None'])
    net2 = AirNet(opt2).to(DEV).eval()
    with torch.no_grad():
        r2, vf2 = net2.R(x_query=q.to(DEV), inter=[0, 0, 0, 0, 0, [0, 0, 0, 0, 0]])
    assert r2.shape == q.shape and len(vf2) == 10 and vf2[0][0][2] == [] and vf2[0][0][0].shape == (128, 128)
