"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's convolutional / ViT plug-ins of the model seam
(net/model.py:17,31): ResNetEncoder (net/encoder_ResNet.py:4-47), DGRN = ResNetDecoder (net/decoder_DGRN.py:9-158) with its
modulated deformable convolution (net/utils/deform_conv.py:10-67) and ViTEncoder (net/encoder_ViT.py:17-203).
Never imported by the product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

Functional PyTorch over a flat {state_dict key: tensor} dict, like oracle/airnet_oracle.py; backward by autograd.

Parity status
  * ResBlock (stride 1 / 2, train mode), ResNetEncoder (eval and train), SFT_layer and ViTEncoder (eval and train, Dropout off) are
    PINNED by goldens produced from the imported reference (tests/golden/make_golden.py convnets / vit;
    tests/test_oracle_convnets.py).  ViT encoder + plain Uformer decoder, the one end-to-end configuration with these plug-ins
    that runs in the reference, is pinned by `model_vit_uformer`.
  * DCN_layer is **parity unpinned**: the reference ends in `assert False` (deform_conv.py:64) because its arithmetic lived in
    the third-party mmcv `modulated_deform_conv2d` (version unpinned -- the reference has no requirements file --, import and
    call commented out at deform_conv.py:7,66-67), absent from the tree and from this image.  `dcn_v2` below follows
    deform_conv.py:56-62 for the offset / mask plumbing and restates the published DCNv2 definition (Zhu et al., "Deformable
    ConvNets v2: More Deformable, Better Results", eq. 1, as mmcv's modulated_deform_conv2d implements it): per output pixel
    p and kernel tap k,
        y(p) = sum_k w_k . m_k . x(p + p_k + dp_k),     x(.) bilinear, zero outside the image,
    offset channels interleaved (dy, dx) per tap, deformable_groups = 1.  It is anchored by known-answer tests (zero offsets
    => sigmoid(0) * conv2d; integer offsets => shifted conv2d; an affine image is sampled exactly).
  * Because DCN_layer asserts, DGM / DGB / DGG / DGRN.forward cannot be executed in the reference either: their wiring
    (decoder_DGRN.py:22-32,73-84,99-110,144-158) is restated here line by line and is **parity unpinned** as an assembly; its
    pinned ingredients are the plain convolutions, LeakyReLU and SFT_layer.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# ResNet encoder  (net/encoder_ResNet.py)
# --------------------------------------------------------------------------------------
def batch_norm2d(st, p, x, training, bn_update=None, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d: batch statistics in train mode (biased variance for the normalisation, unbiased for running_var)."""
    if training:
        if bn_update is not None:
            bn_update[p] = (x.mean((0, 2, 3)).detach(), x.var((0, 2, 3), unbiased=True).detach())
        return F.batch_norm(x, None, None, st[p + 'weight'], st[p + 'bias'], True, momentum, eps)
    return F.batch_norm(x, st[p + 'running_mean'], st[p + 'running_var'], st[p + 'weight'], st[p + 'bias'], False, momentum, eps)


def res_block(st, p, x, stride, training, bn_update=None):
    """ResBlock.forward, encoder_ResNet.py:4-20: LReLU(0.1)(backbone(x) + shortcut(x))."""
    y = F.conv2d(x, st[p + 'backbone.0.weight'], None, stride, 1)
    y = F.leaky_relu(batch_norm2d(st, p + 'backbone.1.', y, training, bn_update), 0.1)
    y = F.conv2d(y, st[p + 'backbone.3.weight'], None, 1, 1)
    y = batch_norm2d(st, p + 'backbone.4.', y, training, bn_update)
    s = F.conv2d(x, st[p + 'shortcut.0.weight'], None, stride, 0)
    s = batch_norm2d(st, p + 'shortcut.1.', s, training, bn_update)
    return F.leaky_relu(y + s, 0.1)


def resnet_encoder(st, p, x, training, bn_update=None):
    """ResNetEncoder.forward, encoder_ResNet.py:42-47 -> (fea [B, dim], [out], inter [B, dim/4, H, W])."""
    inter = res_block(st, p + 'E_pre.', x, 1, training, bn_update)
    y = res_block(st, p + 'E.0.', inter, 2, training, bn_update)
    y = res_block(st, p + 'E.1.', y, 2, training, bn_update)
    fea = y.mean((2, 3))                                       # AdaptiveAvgPool2d(1) + squeeze
    out = F.linear(fea, st[p + 'mlp.0.weight'], st[p + 'mlp.0.bias'])
    out = F.linear(F.leaky_relu(out, 0.1), st[p + 'mlp.2.weight'], st[p + 'mlp.2.bias'])
    return fea, [out], inter


# --------------------------------------------------------------------------------------
# DCNv2 (parity unpinned, see the header) and the DGRN decoder  (net/decoder_DGRN.py)
# --------------------------------------------------------------------------------------
def bilinear_sample(x, py, px):
    """x: [B, C, H, W];  py, px: [B, H, W] float sampling positions -> [B, C, H, W]; zero outside the image
    (each of the four neighbours contributes only where it lies inside)."""
    B, C, H, W = x.shape
    y0, x0 = torch.floor(py), torch.floor(px)
    wy1, wx1 = py - y0, px - x0
    out = 0
    flat = x.reshape(B, C, H * W)
    for dy, wy in ((0, 1 - wy1), (1, wy1)):
        for dx, wx in ((0, 1 - wx1), (1, wx1)):
            yy, xx = y0 + dy, x0 + dx
            ok = ((yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)).to(x.dtype)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().reshape(B, 1, H * W).expand(B, C, H * W)
            v = torch.gather(flat, 2, idx).reshape(B, C, H, W)
            out = out + v * (wy * wx * ok).unsqueeze(1)
    return out


def dcn_v2(x, offset, mask, weight, bias=None):
    """Modulated deformable 3x3 convolution, stride 1, padding 1, dilation 1, groups 1, deformable_groups 1.
    x [B, Cin, H, W];  offset [B, 18, H, W] = (dy, dx) interleaved per tap k = ky*3 + kx;  mask [B, 9, H, W] (already sigmoid-ed);
    weight [Cout, Cin, 3, 3]."""
    B, C, H, W = x.shape
    gy = torch.arange(H, dtype=x.dtype).view(1, H, 1)
    gx = torch.arange(W, dtype=x.dtype).view(1, 1, W)
    out = 0
    for k in range(9):
        ky, kx = k // 3, k % 3
        py = gy + (ky - 1) + offset[:, 2 * k]
        px = gx + (kx - 1) + offset[:, 2 * k + 1]
        col = bilinear_sample(x, py, px) * mask[:, k:k + 1]                       # [B, Cin, H, W]
        out = out + torch.einsum('bchw,oc->bohw', col, weight[:, :, ky, kx])
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def dcn_layer(st, p, x, inter):
    """DCN_layer.forward, deform_conv.py:56-67: offsets and masks from a 3x3 conv on cat[x, inter]; o1, o2, mask = chunk(3);
    offset = cat(o1, o2); mask = sigmoid(mask); then the (missing) modulated deformable convolution."""
    out = F.conv2d(torch.cat([x, inter], 1), st[p + 'conv_offset_mask.weight'], st[p + 'conv_offset_mask.bias'], 1, 1)
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    return dcn_v2(x, offset, torch.sigmoid(mask), st[p + 'weight'], st.get(p + 'bias'))


def sft_layer(st, p, x, inter):
    """SFT_layer.forward, decoder_DGRN.py:49-57: x * gamma(inter) + beta(inter), two 1x1-conv MLPs with LReLU(0.1)."""
    def mlp(q):
        h = F.leaky_relu(F.conv2d(inter, st[p + q + '.0.weight']), 0.1)
        return F.conv2d(h, st[p + q + '.2.weight'])
    return x * mlp('conv_gamma') + mlp('conv_beta')


def dgm(st, p, x, inter, dcn=dcn_layer):
    """DGM.forward, decoder_DGRN.py:22-32: x + DCN(x, inter) + SFT(x, inter)."""
    return x + dcn(st, p + 'dcn.', x, inter) + sft_layer(st, p + 'sft.', x, inter)


def dgb(st, p, x, inter, dcn=dcn_layer):
    """DGB.forward, decoder_DGRN.py:73-84."""
    out = F.leaky_relu(dgm(st, p + 'dgm1.', x, inter, dcn), 0.1)
    out = F.leaky_relu(F.conv2d(out, st[p + 'conv1.weight'], st[p + 'conv1.bias'], 1, 1), 0.1)
    out = F.leaky_relu(dgm(st, p + 'dgm2.', out, inter, dcn), 0.1)
    return F.conv2d(out, st[p + 'conv2.weight'], st[p + 'conv2.bias'], 1, 1) + x


def dgg(st, p, x, inter, n_blocks=5, dcn=dcn_layer):
    """DGG.forward, decoder_DGRN.py:99-110."""
    res = x
    for i in range(n_blocks):
        res = dgb(st, p + f'body.{i}.', res, inter, dcn)
    res = F.conv2d(res, st[p + f'body.{n_blocks}.weight'], st[p + f'body.{n_blocks}.bias'], 1, 1)
    return res + x


def dgrn(st, p, x, inter, n_groups=5, n_blocks=5, dcn=dcn_layer):
    """DGRN.forward, decoder_DGRN.py:144-158 (no global residual to the input image)."""
    x = F.conv2d(x, st[p + 'head.0.weight'], st[p + 'head.0.bias'], 1, 1)
    res = x
    for i in range(n_groups):
        res = dgg(st, p + f'body.{i}.', res, inter, n_blocks, dcn)
    res = F.conv2d(res, st[p + f'body.{n_groups}.weight'], st[p + f'body.{n_groups}.bias'], 1, 1)
    res = res + x
    return F.conv2d(res, st[p + 'tail.0.weight'], st[p + 'tail.0.bias'], 1, 1)


# --------------------------------------------------------------------------------------
# ViT encoder  (net/encoder_ViT.py)
# --------------------------------------------------------------------------------------
def _drop(t, drop, layer, which):
    """nn.Dropout with the product's counter-based mask (oracle/dropout_hash.py); drop = (seed, site base, p) or None."""
    if drop is None or drop[2] <= 0:
        return t
    import dropout_hash as DH
    seed, base, p = drop
    m = torch.from_numpy(DH.keep_mask(seed, DH.vit_site(base, layer, which), tuple(t.shape), p))
    return t * m.to(t.dtype) / (1.0 - p)


def attn_band_masks(decompose_type, n=64):
    """Band masks of the ViT's attention-map decomposition in UN-shifted spectrum coordinates, [nb, n, n] bool
    (encoder_ViT.py:51-60: FrequencyDecompose('frequency_decompose', 1/nb, dim_head, dim_head) | 'frequency_decompose_dc')."""
    import airnet_oracle as A
    if decompose_type == 'DC':
        m = torch.zeros(2, n, n, dtype=torch.bool)
        m[0, 0, 0] = True
        m[1] = ~m[0]
        return m
    nb = int(decompose_type.split('_')[0])
    shifted = torch.stack(A.band_masks('frequency_decompose', 1. / nb, n, n))
    return torch.fft.ifftshift(shifted, dim=(-2, -1))


def vit_attention(st, p, x, heads, decompose_type='none', drop=None, layer=0):
    """Attention.forward, encoder_ViT.py:76-98: softmax(q k^T scale) [+ sum_i lamb_i band_i(attn)] -> dropout -> attn v -> to_out -> dropout."""
    B, N, C = x.shape
    qkv = F.linear(x, st[p + 'to_qkv.weight']).chunk(3, dim=-1)
    q, k, v = (t.reshape(B, N, heads, -1).transpose(1, 2) for t in qkv)
    D = q.shape[-1]
    attn = ((q @ k.transpose(-1, -2)) * D ** -0.5).softmax(-1)
    if decompose_type != 'none':
        masks = attn_band_masks(decompose_type, D)                               # sized dim_head (:56,60): needs N == D
        spec = torch.fft.fft2(attn)
        bands = torch.stack([torch.fft.ifft2(spec * m).real for m in masks.to(attn.dtype)], 0)   # [nb, B, heads, N, N]
        attn = attn + (bands * st[p + 'lamb'][:, :, :, None, None]).sum(0)
    attn = _drop(attn, drop, layer, 'attn')
    out = (attn @ v).transpose(1, 2).reshape(B, N, heads * D)
    return _drop(F.linear(out, st[p + 'to_out.0.weight'], st[p + 'to_out.0.bias']), drop, layer, 'out')


def vit_encoder(st, p, opt, x, training, bn_update=None, depth=12, heads=12, patch=16, drop=None):
    """ViTEncoder.forward, encoder_ViT.py:181-203.  drop = (seed, site base, p): the train-mode Dropouts with the product's masks;
    None: identity (eval, or the p = 0 golden runs).  -> (fea [B, encoder_dim], [out], inter [B, encoder_dim, H, W])."""
    B, C, H, W = x.shape
    hh, ww = H // patch, W // patch
    dtype = getattr(opt, 'frequency_decompose_type', 'none') or 'none'
    t = x.reshape(B, C, hh, patch, ww, patch).permute(0, 2, 4, 3, 5, 1).reshape(B, hh * ww, patch * patch * C)   # b (h w) (p1 p2 c)
    e = p + 'to_patch_embedding.'
    t = F.layer_norm(t, (t.shape[-1],), st[e + '1.weight'], st[e + '1.bias'])
    t = F.linear(t, st[e + '2.weight'], st[e + '2.bias'])
    t = F.layer_norm(t, (t.shape[-1],), st[e + '3.weight'], st[e + '3.bias'])
    dim = t.shape[-1]
    t = _drop(t + st[p + 'pos_embedding'][:, :t.shape[1]], drop, 0, 'emb')
    for i in range(depth):
        a = p + f'transformer.layers.{i}.0.'
        t = vit_attention(st, a + 'fn.', F.layer_norm(t, (dim,), st[a + 'norm.weight'], st[a + 'norm.bias']), heads, dtype, drop, i) + t
        f = p + f'transformer.layers.{i}.1.'
        h = F.layer_norm(t, (dim,), st[f + 'norm.weight'], st[f + 'norm.bias'])
        h = _drop(F.gelu(F.linear(h, st[f + 'fn.net.0.weight'], st[f + 'fn.net.0.bias'])), drop, i, 'hidden')
        t = _drop(F.linear(h, st[f + 'fn.net.3.weight'], st[f + 'fn.net.3.bias']), drop, i, 'ff') + t
    t = F.layer_norm(t, (dim,), st[p + 'mlp_head.0.weight'], st[p + 'mlp_head.0.bias'])
    t = F.linear(t, st[p + 'mlp_head.1.weight'], st[p + 'mlp_head.1.bias'])
    inter = t.reshape(-1, opt.encoder_dim, H, W)
    inter = F.leaky_relu(batch_norm2d(st, p + 'norm.0.', inter, training, bn_update), 0.1)
    fea = inter.mean((2, 3))
    out = F.linear(fea, st[p + 'mlp.0.weight'], st[p + 'mlp.0.bias'])
    out = F.linear(F.leaky_relu(out, 0.1), st[p + 'mlp.2.weight'], st[p + 'mlp.2.bias'])
    return fea, [out], inter


# --------------------------------------------------------------------------------------
# AirNet with these plug-ins (net/model.py:59-71, net/utils/moco.py:115-166)
# --------------------------------------------------------------------------------------
def airnet_forward(st, opt, x_query, x_key, training, decoder, update_state=True, drop=None):
    """AirNet.forward for the ResNet / ViT encoders.  The reference's MoCo indexes `range(opt.L)` heads on their 1-element output
    list and fails (moco.py:127-128); as the build does, the contrastive loss runs over len(q) = 1 head against
    queue[0].  decoder(st, x_query, inter) -> restored.  Train: (restored, [logits], [labels]); eval: restored."""
    import airnet_oracle as A
    def vit(s, p, x, t, b=None):
        d = None
        if drop is not None and t:                       # drop = (seed, p): train-mode Dropout masks of the product, per encoder copy
            import dropout_hash as DH
            d = (drop[0], DH.site_base(p), drop[1])
        return vit_encoder(s, p, opt, x, t, b, drop=d)
    enc = resnet_encoder if opt.encoder_type == 'ResNet' else vit
    if not training:
        _, _, inter = enc(st, 'E.E.encoder_q.', x_query, False)
        return decoder(st, x_query, inter)
    bn, bnk = {}, {}
    _, q, inter = enc(st, 'E.E.encoder_q.', x_query, True, bn)
    q = [F.normalize(t, dim=1) for t in q]
    with torch.no_grad():
        if update_state:
            A.moco_momentum_update(st)
        kst = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st.items()}
        _, k, _ = enc(kst, 'E.E.encoder_k.', x_key, True, bnk)
        k = [F.normalize(t, dim=1) for t in k]
    queue = st['E.E.queue']
    logits, labels = [], []
    for i in range(len(q)):
        l_pos = torch.einsum('nc,nc->n', q[i], k[i]).unsqueeze(-1)
        l_neg = torch.einsum('nc,ck->nk', q[i], queue[i].clone().detach())
        logits.append(torch.cat([l_pos, l_neg], 1) / A.MOCO_T)
        labels.append(torch.zeros(logits[i].shape[0], dtype=torch.long))
    if update_state:
        with torch.no_grad():
            bsz, ptr, K = k[0].shape[0], int(st['E.E.queue_ptr']), queue.shape[2]
            for i in range(len(q)):
                queue[i][:, ptr:ptr + bsz] = k[i].transpose(0, 1)
            st['E.E.queue_ptr'][0] = (ptr + bsz) % K
            for d_ in (bn, bnk):
                for n, (m, v) in d_.items():
                    st[n + 'running_mean'] = st[n + 'running_mean'] * 0.9 + 0.1 * m
                    st[n + 'running_var'] = st[n + 'running_var'] * 0.9 + 0.1 * v
    return decoder(st, x_query, inter), logits, labels
