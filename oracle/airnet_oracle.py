"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference hot path.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``frequency-wised_all-in-one_image_restoration_model_amd/``)
never imports anything from ``oracle/`` and fails loudly without its HIP library.

It restates, in plain functional PyTorch (CPU, fp32 or fp64), the forward math of
``AirNet`` for the configurations of the reference that actually run
(SURVEY.md section 0.1): Uformer encoder (``freq`` or ``origin`` MSA) + Uformer
decoder with learned frequency selection (``all_<k>_bands`` / ``all_DC``).
Backward is obtained by autograd over this restatement.  All citations are
``file:line`` relative to the reference root.

Parity is PINNED: ``tests/golden/make_golden.py`` imports the real reference in
the build container and stores its outputs for name-seeded weights;
``tests/test_oracle_golden.py`` checks this restatement against those vectors.

Everything operates on a flat ``state`` dict {state_dict key: tensor} with the
reference's key names, so no module classes are shared
with either the reference or the product.
"""
import math
import zlib
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

WIN = 8                      # decoder_Uformer.py:838 / encoder_Uformer.py:749  win_size=8
DEC_DEPTHS = [2, 2, 8, 8, 2, 8, 8, 2, 2]      # decoder_Uformer.py:837
ENC_DEPTHS = [2, 2, 2, 2, 2]                  # encoder_Uformer.py:748 (first five are used)
NUM_HEADS = [1, 2, 4, 8, 16, 16, 8, 4, 2]     # decoder_Uformer.py:837 / encoder_Uformer.py:748
DROP_PATH_RATE = 0.1                          # decoder_Uformer.py:839 / encoder_Uformer.py:750
MOCO_M = 0.999                                # net/utils/moco.py:16
MOCO_T = 0.07                                 # net/utils/moco.py:16


def make_opt(**kw):
    """The subset of ``option.options`` the hot path reads, with option.py defaults."""
    d = dict(L=3, encoder_dim=256, encoder_embed_dim=28, embed_dim=56,
             degradation_embedding_method=['all_3_bands'], encoder_msa_type='freq',
             batch_size=2, patch_size=128, contrast_loss_weight=0.6,
             encoder_type='Uformer', decoder_type='Uformer', debug_mode=False,
             frequency_decompose_type='none', learnable_modulator=False)
    d.update(kw)
    return SimpleNamespace(**d)


# --------------------------------------------------------------------------------------
# Frequency decomposition  (net/utils/frequency_decompose.py)
# --------------------------------------------------------------------------------------
def band_masks(kind, size, h, w):
    """Boolean band masks in fftshift-ed coordinates.

    frequency_decompose.py:17-26 (grid, centre (int(w/2), int(h/2)), max_radius),
    :37-49 (``frequency_decompose``: [0,s) ... [1-s,1]) and :79-88
    (``frequency_decompose_1``: DC, (0,s] ... (1-s,1]).  All arithmetic is done in the
    same dtypes as the reference (int64 grid -> float32 sqrt) so the <=/< edge cases
    (e.g. bins with fx^2+fy^2 == r^2) fall on the same side.
    """
    Y = torch.arange(h).unsqueeze(1)
    X = torch.arange(w).unsqueeze(0)
    num_bands = math.floor(1. / size + 0.1)
    center = torch.tensor([int(w / 2), int(h / 2)])
    dist = torch.sqrt((X - center[0]) ** 2 + (Y - center[1]) ** 2)
    max_radius = torch.sqrt(center[0] ** 2 + center[1] ** 2)
    last = torch.zeros((h, w), dtype=torch.bool)
    out = []
    if kind == 'frequency_decompose':
        for sz in torch.linspace(size, 1, num_bands):
            radius = max_radius * sz
            mask = (dist <= radius) if sz == 1.0 else (dist < radius)
            out.append(mask ^ last)
            last = mask
    elif kind == 'frequency_decompose_1':
        for sz in torch.linspace(0, 1, num_bands + 1):
            radius = max_radius * sz
            mask = dist <= radius
            out.append(mask ^ last)
            last = mask
    else:
        raise ValueError(kind)
    return out


def frequency_decompose(x, kind, size, h, w, inverse=True):
    """``FrequencyDecompose.forward`` (frequency_decompose.py:28-125).

    x: [B, C, h, w] real.  Returns [num_bands(+1), B, C, h, w] (or [..., 2] when
    ``inverse is False``; magnitude spectrum when ``inverse == 'visual'``).
    """
    if kind == 'frequency_decompose_dc':                      # :109-118
        n = x.shape[2]
        x_d = x.mean(-1, keepdim=True).mean(-2, keepdim=True).repeat(1, 1, n, n)
        return torch.stack([x_d, x - x_d], 0)
    fre = torch.fft.fftshift(torch.fft.fft2(x))               # :32 / :74  (shift over ALL dims)
    outs = []
    for m in band_masks(kind, size, h, w):
        d = m.to(x.device)[None, None] * fre                   # :51 / :90
        if isinstance(inverse, str) and inverse == 'visual':  # :54-55
            d = d.abs()
        elif inverse is True:                                  # :56-58
            d = torch.fft.ifft2(torch.fft.ifftshift(d)).real
        elif inverse is False:                                 # :59-61
            d = torch.fft.ifftshift(d)
            d = torch.stack((d.real, d.imag), -1)
        else:
            raise AssertionError
        outs.append(d)
    return torch.stack(outs, 0)


# --------------------------------------------------------------------------------------
# Window helpers  (decoder_Uformer.py:387-409, encoder_Uformer.py:398-420)
# --------------------------------------------------------------------------------------
def window_partition(x, win):
    B, H, W, C = x.shape
    x = x.view(B, H // win, win, W // win, win, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, win, win, C)


def window_reverse(windows, win, H, W):
    B = int(windows.shape[0] / (H * W / win / win))
    x = windows.view(B, H // win, W // win, win, win, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def relative_position_index(win):
    """decoder_Uformer.py:201-210 -- [win*win, win*win] int64 index into the (2w-1)^2 table."""
    ch = torch.arange(win)
    coords = torch.stack(torch.meshgrid([ch, ch], indexing='ij'))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += win - 1
    rel[:, :, 1] += win - 1
    rel[:, :, 0] *= 2 * win - 1
    return rel.sum(-1)


def shift_attn_mask(H, W, win, shift, dtype=torch.float32):
    """SW-MSA mask, decoder_Uformer.py:634-651 -- [nW, win*win, win*win] of 0 / -100."""
    m = torch.zeros((1, H, W, 1), dtype=dtype)
    sl = (slice(0, -win), slice(-win, -shift), slice(-shift, None))
    cnt = 0
    for hs in sl:
        for ws in sl:
            m[:, hs, ws, :] = cnt
            cnt += 1
    mw = window_partition(m, win).view(-1, win * win)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def gelu(x):
    return F.gelu(x)                                           # nn.GELU() default = erf form


# --------------------------------------------------------------------------------------
# LeFF  (net/utils/leff.py:92-117)
# --------------------------------------------------------------------------------------
def leff(st, p, x):
    bs, hw, c = x.shape
    hh = int(math.sqrt(hw))
    x = gelu(F.linear(x, st[p + 'linear1.0.weight'], st[p + 'linear1.0.bias']))
    x = x.view(bs, hh, hh, -1).permute(0, 3, 1, 2)
    x = gelu(F.conv2d(x, st[p + 'conv.0.weight'], st[p + 'conv.0.bias'], padding=1,
                      groups=x.shape[1]))
    x = x.permute(0, 2, 3, 1).reshape(bs, hw, -1)
    return F.linear(x, st[p + 'linear2.0.weight'], st[p + 'linear2.0.bias'])


def linear_projection(st, p, x, heads):
    """LinearProjection.forward, decoder_Uformer.py:98-125 / encoder_Uformer.py:89-100."""
    B_, N, C = x.shape
    q = F.linear(x, st[p + 'to_q.weight'], st[p + 'to_q.bias'])
    kv = F.linear(x, st[p + 'to_kv.weight'], st[p + 'to_kv.bias'])
    q = q.reshape(B_, N, 1, heads, C // heads).permute(2, 0, 3, 1, 4)[0]
    kv = kv.reshape(B_, N, 2, heads, C // heads).permute(2, 0, 3, 1, 4)
    return q, kv[0], kv[1]


def rel_bias(table, win):
    """decoder_Uformer.py:245-247: gather [225,h] by the index -> [h, N, N]."""
    idx = relative_position_index(win).view(-1)
    n = win * win
    return table[idx].view(n, n, -1).permute(2, 0, 1).contiguous()


# --------------------------------------------------------------------------------------
# Decoder window attention with learned frequency selection
# (decoder_Uformer.py:235-299)
# --------------------------------------------------------------------------------------
def lfs_config(methods):
    """decoder_Uformer.py:166-174 -> (num_bands, decompose kind, size) or None."""
    cfg = None
    for t in methods:
        if 'all' not in t:
            continue
        if t.split('_')[-1] == 'bands':
            nb = int(t.split('_')[-2])
            cfg = (nb, 'frequency_decompose_1', 1. / (nb - 1))
        elif t.split('_')[-1] == 'DC':
            cfg = (2, 'frequency_decompose_dc', 0.5)
    return cfg


def lfs_lambda(st, p, i, inter_i):
    """decoder_Uformer.py:280-284: LN(448) -> Linear(448,h) -> mean over tokens -> MLP."""
    e = F.layer_norm(inter_i, (inter_i.shape[-1],), st[p + f'mlp_head.{i}.0.weight'],
                     st[p + f'mlp_head.{i}.0.bias'])
    e = F.linear(e, st[p + f'mlp_head.{i}.1.weight'], st[p + f'mlp_head.{i}.1.bias'])
    e = e.mean(1, keepdim=True)                                 # AdaptiveAvgPool1d(1) over tokens
    e = F.linear(e, st[p + f'mlp.{i}.0.weight'], st[p + f'mlp.{i}.0.bias'])
    e = F.leaky_relu(e, 0.1)
    return F.linear(e, st[p + f'mlp.{i}.2.weight'], st[p + f'mlp.{i}.2.bias'])   # [B,1,h]


def window_attention_lfs(st, p, x, heads, num_win, all_inter, mask, lfs, want_attn=False):
    B_, N, C = x.shape
    q, k, v = linear_projection(st, p + 'qkv.', x, heads)
    scale = (C // heads) ** -0.5
    attn = (q * scale) @ k.transpose(-2, -1)
    attn = attn + rel_bias(st[p + 'relative_position_bias_table'], WIN).unsqueeze(0)
    if mask is not None:                                        # :253-258
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = attn.softmax(-1)
    lambs = []
    if lfs is not None:                                         # :275-288
        nb, kind, size = lfs
        bands = frequency_decompose(attn, kind, size, N, N, True)
        for i in range(1, nb):
            lam = lfs_lambda(st, p, i, all_inter[i])            # [B,1,h]
            lambs.append(lam)
            band = bands[i].view(-1, num_win, heads, N, N) * lam[:, :, :, None, None]
            attn = attn + band.view(-1, heads, N, N)
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    out = F.linear(out, st[p + 'proj.weight'], st[p + 'proj.bias'])
    if want_attn:
        return out, attn, lambs
    return out


def drop_path_apply(x, dp):
    """dp: None (identity) or a per-sample scale tensor [B] (mask / keep_prob)."""
    if dp is None:
        return x
    return x * dp.view(-1, *([1] * (x.ndim - 1))).to(x.dtype)


def drop_path_pair(dp):
    """A block calls self.drop_path twice (decoder_Uformer.py:739,751 / encoder_Uformer.py:679-680): timm draws an
    INDEPENDENT mask for the attention branch and for the MLP branch.  dp: None | (dp_attn, dp_mlp)."""
    if dp is None:
        return None, None
    assert isinstance(dp, (tuple, list)) and len(dp) == 2, 'dp = (attention-branch scale, MLP-branch scale)'
    return dp


def visual_spectrum(tokens, H, W):
    """decoder_Uformer.py:668-673 / :731-736 (debug_mode): |fftshift(fft2(.))| of the [B, C, H, W] map -- the single band of
    FrequencyDecompose('frequency_decompose', 1, H, W, inverse='visual') -- averaged over batch and channels -> [H, W].
    (The reference's fftshift also rolls the batch / channel axes, frequency_decompose.py:32: the means do not see it.)"""
    B, _, C = tokens.shape
    img = tokens.view(B, H, W, C).permute(0, 3, 1, 2)
    spec = torch.fft.fftshift(torch.fft.fft2(img), dim=(-2, -1)).abs()
    return spec.mean(0).mean(0)


def lewin_block_dec(st, p, x, heads, shift, all_inter, lfs, dp=None, debug=None):
    """LeWinTransformerBlock.forward, decoder_Uformer.py:618-756 (plain + all_* path).
    debug: a list that receives [spectrum_before, spectrum_after, embed_lamb] (`opt.debug_mode`, :753-754)."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    win = min(WIN, H)
    if H <= WIN:
        shift = 0                                              # :531-533
    mask = shift_attn_mask(H, W, win, shift, x.dtype) if shift > 0 else None
    shortcut = x
    y = F.layer_norm(x, (C,), st[p + 'norm1.weight'], st[p + 'norm1.bias']).view(B, H, W, C)
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    if debug is not None:
        before = visual_spectrum(y.reshape(B, H * W, C), H, W)
    yw = window_partition(y, win).view(-1, win * win, C)
    num_win = (H // win) * (W // win)
    aw, _, lambs = window_attention_lfs(st, p + 'attn.', yw, heads, num_win, all_inter, mask, lfs, want_attn=True)
    y = window_reverse(aw.view(-1, win, win, C), win, H, W)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    if debug is not None:                                      # embed_lamb = the LAST band's lambda (:277-296), [] without LFS
        debug.append([before, visual_spectrum(y.reshape(B, H * W, C), H, W), lambs[-1] if lambs else []])
    dp_attn, dp_mlp = drop_path_pair(dp)
    x = shortcut + drop_path_apply(y.view(B, H * W, C), dp_attn)
    z = leff(st, p + 'mlp.', F.layer_norm(x, (C,), st[p + 'norm2.weight'], st[p + 'norm2.bias']))
    return x + drop_path_apply(z, dp_mlp)


def tokens_to_img(x):
    B, L, C = x.shape
    H = int(math.sqrt(L))
    return x.transpose(1, 2).contiguous().view(B, C, H, H)


def downsample(st, p, x):                                       # decoder_Uformer.py:423-430
    o = F.conv2d(tokens_to_img(x), st[p + 'conv.0.weight'], st[p + 'conv.0.bias'], stride=2,
                 padding=1)
    return o.flatten(2).transpose(1, 2).contiguous()


def upsample(st, p, x):                                         # decoder_Uformer.py:443-449
    o = F.conv_transpose2d(tokens_to_img(x), st[p + 'deconv.0.weight'], st[p + 'deconv.0.bias'],
                           stride=2)
    return o.flatten(2).transpose(1, 2).contiguous()


def input_proj(st, p, x):                                       # decoder_Uformer.py:467-472
    o = F.leaky_relu(F.conv2d(x, st[p + 'proj.0.weight'], st[p + 'proj.0.bias'], padding=1), 0.01)
    return o.flatten(2).transpose(1, 2).contiguous()


def dpr_lists(depths):
    """Stochastic-depth rates, decoder_Uformer.py:876-878 / encoder_Uformer.py:781-783."""
    enc = [x.item() for x in torch.linspace(0, DROP_PATH_RATE, sum(depths[:4]))]
    conv = [DROP_PATH_RATE] * depths[4]
    return enc, conv, enc[::-1]


def decoder_layer_table(opt):
    """(name, dim multiplier, heads, depth, per-block drop-path rates) in forward order."""
    enc, conv, dec = dpr_lists(DEC_DEPTHS)
    d = DEC_DEPTHS
    t = []
    o = 0
    for i in range(4):
        t.append((f'encoderlayer_{i}', 2 ** i, NUM_HEADS[i], d[i], enc[o:o + d[i]]))
        o += d[i]
    t.append(('bottleneck_0', 16, NUM_HEADS[4], d[4], conv))
    t.append(('bottleneck_1', 16, NUM_HEADS[4], d[4], conv))
    t.append(('decoderlayer_3', 16, NUM_HEADS[5], d[5], dec[:d[5]]))
    t.append(('decoderlayer_2', 8, NUM_HEADS[6], d[6], dec[d[5]:d[5] + d[6]]))
    t.append(('decoderlayer_1', 4, NUM_HEADS[7], d[7], dec[d[5] + d[6]:d[5] + d[6] + d[7]]))
    t.append(('decoderlayer_0', 2, NUM_HEADS[8], d[8], dec[sum(d[5:8]):sum(d[5:9])]))
    return t


def run_layer(st, p, x, heads, depth, block_fn, dps, **kw):
    for b in range(depth):
        shift = 0 if b % 2 == 0 else WIN // 2
        x = block_fn(st, f'{p}blocks.{b}.', x, heads, shift, dp=None if dps is None else dps.get(f'{p}blocks.{b}.'), **kw)
    return x


def uformer_decoder(st, p, opt, x, inter, dps=None):
    """UformerDecoder.forward, decoder_Uformer.py:1117-1171 (all_* / plain path).  With `opt.debug_mode` the return value is
    (restored, visual_freqs), visual_freqs[layer][block] = [spectrum_before, spectrum_after, embed_lamb] (:1168-1169)."""
    lfs = lfs_config(opt.degradation_embedding_method)
    tab = {n: (h, d) for n, _, h, d, _ in decoder_layer_table(opt)}
    kw = dict(all_inter=inter, lfs=lfs)
    visual = [] if getattr(opt, 'debug_mode', False) else None

    def layer(name, y):
        h, d = tab[name]
        if visual is not None:
            visual.append([])
            return run_layer(st, f'{p}{name}.', y, h, d, lewin_block_dec, dps, debug=visual[-1], **kw)
        return run_layer(st, f'{p}{name}.', y, h, d, lewin_block_dec, dps, **kw)

    y = input_proj(st, p + 'input_proj.', x)
    conv = []
    for i in range(4):
        y = layer(f'encoderlayer_{i}', y)
        conv.append(y)
        y = downsample(st, p + f'dowsample_{i}.', y)
    y = layer('bottleneck_0', y)
    y = layer('bottleneck_1', y)
    for i in reversed(range(4)):
        y = upsample(st, p + f'upsample_{i}.', y)
        y = torch.cat([y, conv[i]], -1)
        y = layer(f'decoderlayer_{i}', y)
    B, L, C = y.shape
    H = int(math.sqrt(L))
    y = F.conv2d(y.transpose(1, 2).view(B, C, H, H), st[p + 'output_proj.proj.0.weight'],
                 st[p + 'output_proj.proj.0.bias'], padding=1)
    if visual is not None:
        return x + y, visual
    return x + y


# --------------------------------------------------------------------------------------
# Encoder: frequency window attention (encoder_Uformer.py:256-310) and plain W-MSA (:152-183)
# --------------------------------------------------------------------------------------
def freq_window_attention(st, p, x, heads, L, kind, mask):
    B_, N, C = x.shape                                          # (l b nw) token dim
    q, k, v = linear_projection(st, p + 'qkv.', x, heads)
    d = C // heads

    def regroup(t):                                             # '(l bnw) h token d -> bnw h (l token) d'
        return t.view(L, B_ // L, heads, N, d).permute(1, 2, 0, 3, 4).reshape(B_ // L, heads, L * N, d)

    q, k, v = regroup(q), regroup(k), regroup(v)
    attn = (q * d ** -0.5) @ k.transpose(-2, -1)
    bias = torch.stack([rel_bias(st[p + f'relative_position_bias_table.{i}'], WIN) for i in range(L * L)], 0)
    bias = bias.view(L, L, heads, N, N).permute(2, 0, 3, 1, 4).reshape(1, heads, L * N, L * N)   # :275-276
    attn = attn + bias
    eye = torch.eye(L, dtype=x.dtype)
    mf = (1 - eye) * -100.0 if kind == 'intra' else eye * -100.0    # :246-249
    attn = attn + mf.repeat_interleave(N, 0).repeat_interleave(N, 1)[None, None]
    if mask is not None:                                        # :286-291
        nW = mask.shape[0]
        m = mask.repeat(1, L, L)
        attn = attn.view(B_ // L // nW, nW, heads, L * N, L * N) + m.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, L * N, L * N)
    attn = attn.softmax(-1)
    o = attn @ v                                                # bnw h (l token) d
    o = o.view(B_ // L, heads, L, N, d).permute(2, 0, 1, 3, 4).reshape(B_, heads, N, d)
    o = o.transpose(1, 2).reshape(B_, N, C)
    return F.linear(o, st[p + 'proj.weight'], st[p + 'proj.bias'])


def window_attention_plain(st, p, x, heads, mask):
    B_, N, C = x.shape
    q, k, v = linear_projection(st, p + 'qkv.', x, heads)
    attn = (q * (C // heads) ** -0.5) @ k.transpose(-2, -1)
    attn = attn + rel_bias(st[p + 'relative_position_bias_table'], WIN).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = attn.softmax(-1)
    o = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(o, st[p + 'proj.weight'], st[p + 'proj.bias'])


def lewin_block_enc(st, p, x, heads, shift, L, msa, dp=None):
    """LeWinTransformerBlock.forward, encoder_Uformer.py:597-682."""
    B, Ltok, C = x.shape
    H = W = int(math.sqrt(Ltok))
    win = min(WIN, H)
    if H <= WIN:
        shift = 0
    mask = shift_attn_mask(H, W, win, shift, x.dtype) if shift > 0 else None
    shortcut = x
    y = F.layer_norm(x, (C,), st[p + 'norm1.weight'], st[p + 'norm1.bias']).view(B, H, W, C)
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    yw = window_partition(y, win).view(-1, win * win, C)
    if msa == 'origin':
        aw = window_attention_plain(st, p + 'attn.', yw, heads, mask)
    else:
        aw = freq_window_attention(st, p + 'attn_intra.', yw, heads, L, 'intra', mask)
        aw = freq_window_attention(st, p + 'attn_inter.', aw, heads, L, 'inter', mask)
    y = window_reverse(aw.view(-1, win, win, C), win, H, W)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    dp_attn, dp_mlp = drop_path_pair(dp)
    x = shortcut + drop_path_apply(y.view(B, H * W, C), dp_attn)
    z = leff(st, p + 'mlp.', F.layer_norm(x, (C,), st[p + 'norm2.weight'], st[p + 'norm2.bias']))
    return x + drop_path_apply(z, dp_mlp)


def encoder_layer_table():
    enc, conv, _ = dpr_lists(ENC_DEPTHS + [2, 2, 2, 2])
    t, o = [], 0
    for i in range(4):
        t.append((f'encoderlayer_{i}', 2 ** i, NUM_HEADS[i], 2, enc[o:o + 2]))
        o += 2
    t.append(('conv', 16, NUM_HEADS[4], 2, conv))
    return t


def uformer_encoder(st, p, opt, x, training, dps=None, bn_update=None):
    """UformerEncoder.forward, encoder_Uformer.py:959-986.  Returns (None, out list, inter tuple).

    ``bn_update``: optional dict that receives the batch statistics the train-mode
    BatchNorm2d would fold into running_mean / running_var (momentum 0.1).
    """
    L = opt.L
    B = x.shape[0]
    img = x.shape[-1]
    if L != 1:                                                  # :964-966
        x = frequency_decompose(x, 'frequency_decompose_1', 1. / (L - 1), img, img, True)
        x = x.reshape(L * B, *x.shape[2:])
    u = p + 'uformer.'
    y = input_proj(st, u + 'input_proj.', x)
    kw = dict(L=L, msa=opt.encoder_msa_type)
    for i in range(4):
        y = run_layer(st, u + f'encoderlayer_{i}.', y, NUM_HEADS[i], 2, lewin_block_enc, dps, **kw)
        y = downsample(st, u + f'dowsample_{i}.', y)
    y = run_layer(st, u + 'conv.', y, NUM_HEADS[4], 2, lewin_block_enc, dps, **kw)
    y = y.view(L, B, y.shape[1], y.shape[2])
    inter = tuple(y.unbind(0))
    out = []
    for i in range(L):                                          # :975-984
        f = F.layer_norm(inter[i], (inter[i].shape[-1],), st[p + f'mlp_head.{i}.0.weight'],
                         st[p + f'mlp_head.{i}.0.bias'])
        f = F.linear(f, st[p + f'mlp_head.{i}.1.weight'], st[p + f'mlp_head.{i}.1.bias'])
        f = f.reshape(f.shape[0], opt.encoder_dim, img, img)
        n = p + f'norm.{i}.0.'
        if training:
            if bn_update is not None:
                m = f.mean((0, 2, 3))
                v = f.var((0, 2, 3), unbiased=True)
                bn_update[n] = (m.detach(), v.detach())
            f = F.batch_norm(f, None, None, st[n + 'weight'], st[n + 'bias'], True, 0.1, 1e-5)
        else:
            f = F.batch_norm(f, st[n + 'running_mean'], st[n + 'running_var'], st[n + 'weight'],
                             st[n + 'bias'], False, 0.1, 1e-5)
        f = F.leaky_relu(f, 0.1).mean((2, 3))
        f = F.linear(f, st[p + f'mlp.{i}.0.weight'], st[p + f'mlp.{i}.0.bias'])
        f = F.linear(F.leaky_relu(f, 0.1), st[p + f'mlp.{i}.2.weight'], st[p + f'mlp.{i}.2.bias'])
        out.append(f)
    return None, out, inter


# --------------------------------------------------------------------------------------
# MoCo + AirNet  (net/utils/moco.py:115-166, net/model.py:59-71)
# --------------------------------------------------------------------------------------
def moco_momentum_update(st, m=MOCO_M):
    """moco.py:44-50 over parameters only (buffers such as BN running stats are not touched)."""
    with torch.no_grad():
        for k in list(st.keys()):
            if k.startswith('E.E.encoder_q.') and is_parameter_key(k):
                kk = 'E.E.encoder_k.' + k[len('E.E.encoder_q.'):]
                st[kk] = st[kk] * m + st[k].detach() * (1. - m)


def is_parameter_key(k):
    tail = k.rsplit('.', 1)[-1]
    return tail not in ('relative_position_index', 'mask_freq', 'running_mean', 'running_var',
                        'num_batches_tracked', 'queue', 'queue_ptr')


def airnet_forward(st, opt, x_query, x_key, training, dps=None, update_state=True):
    """AirNet.forward (model.py:59-71) + Encoder (:37-46) + MoCo.forward (moco.py:115-166).

    Train: returns (restored, logits list, labels list); mutates ``st`` in place exactly as the
    reference mutates its buffers (EMA of encoder_k, queue, queue_ptr) when ``update_state``.
    Eval: returns restored.
    """
    if not training:
        _, _, inter = uformer_encoder(st, 'E.E.encoder_q.', opt, x_query, False)
        return uformer_decoder(st, 'R.R.', opt, x_query, inter)
    L = opt.L
    bn = {}
    _, q, inter = uformer_encoder(st, 'E.E.encoder_q.', opt, x_query, True, dps, bn)
    q = [F.normalize(t, dim=1) for t in q]
    with torch.no_grad():
        if update_state:
            moco_momentum_update(st)
        kst = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st.items()}
        bnk = {}
        _, k, _ = uformer_encoder(kst, 'E.E.encoder_k.', opt, x_key, True, dps, bnk)
        k = [F.normalize(t, dim=1) for t in k]
    queue = st['E.E.queue']
    logits, labels = [], []
    for i in range(L):
        l_pos = torch.einsum('nc,nc->n', q[i], k[i]).unsqueeze(-1)
        l_neg = torch.einsum('nc,ck->nk', q[i], queue[i].clone().detach())
        logits.append(torch.cat([l_pos, l_neg], 1) / MOCO_T)
        labels.append(torch.zeros(logits[i].shape[0], dtype=torch.long))
    if update_state:
        with torch.no_grad():
            bsz = k[0].shape[0]
            ptr = int(st['E.E.queue_ptr'])
            K = queue.shape[2]
            assert K % bsz == 0
            for i in range(L):
                queue[i][:, ptr:ptr + bsz] = k[i].transpose(0, 1)
            st['E.E.queue_ptr'][0] = (ptr + bsz) % K
            for d_ in (bn, bnk):
                for n, (m, v) in d_.items():
                    st[n + 'running_mean'] = st[n + 'running_mean'] * 0.9 + 0.1 * m
                    st[n + 'running_var'] = st[n + 'running_var'] * 0.9 + 0.1 * v
                    st[n + 'num_batches_tracked'] = st[n + 'num_batches_tracked'] + 1
    restored = uformer_decoder(st, 'R.R.', opt, x_query, inter, dps)
    return restored, logits, labels


def training_loss(opt, restored, logits, labels, clean):
    """train.py:87-92 (phase-2 objective without the optional frequency-L1 term)."""
    contrast = sum(F.cross_entropy(logits[i], labels[i]) for i in range(len(logits))) / len(logits)
    l1 = F.l1_loss(restored, clean)
    return l1 + opt.contrast_loss_weight * contrast, l1, contrast


def psnr(restored, clean):
    """utils/val_utils.py:52-63: per-image PSNR on [0,1]-clipped tensors, data_range 1."""
    a = restored.clamp(0, 1).double()
    b = clean.clamp(0, 1).double()
    mse = ((a - b) ** 2).flatten(1).mean(1)
    return (10.0 * torch.log10(1.0 / mse)).mean().item()


# --------------------------------------------------------------------------------------
# Name-seeded weights -- build-side convention, no reference code.
# --------------------------------------------------------------------------------------
def seeded_tensor(name, shape, dtype=torch.float32):
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    tail = name.rsplit('.', 1)[-1]
    g = rs.standard_normal(tuple(shape)).astype(np.float32) if len(shape) else np.float32(rs.standard_normal())
    g = torch.from_numpy(np.asarray(g, dtype=np.float32)).reshape(tuple(shape))
    is_norm = tail == 'weight' and len(shape) == 1      # LN / BN scale vectors are the only 1-D weights
    if tail == 'running_var':
        t = 1 + g.abs() * 0.02
    elif tail == 'weight' and is_norm:
        t = 1 + 0.02 * g
    else:
        t = 0.02 * g
    return t.to(dtype)


def fill_state_seeded(schema, dtype=torch.float32):
    """schema: list of (name, shape, dtype str).  Integer buffers are rebuilt, floats seeded.
    encoder_k parameters are copies of encoder_q (moco.py:33-35)."""
    st = {}
    for name, shape, dt in schema:
        tail = name.rsplit('.', 1)[-1]
        if tail == 'relative_position_index':
            st[name] = relative_position_index(WIN)
        elif tail == 'queue_ptr':
            st[name] = torch.zeros(1, dtype=torch.long)
        elif tail == 'num_batches_tracked':
            st[name] = torch.zeros((), dtype=torch.long)
        elif tail == 'mask_freq':
            st[name] = None      # rebuilt analytically inside freq_window_attention
        elif tail == 'queue':
            g = seeded_tensor(name, shape, dtype) / 0.02
            st[name] = F.normalize(g, dim=1)
        else:
            st[name] = seeded_tensor(name, shape, dtype)
    for name in list(st.keys()):
        if name.startswith('E.E.encoder_k.') and is_parameter_key(name):
            st[name] = st['E.E.encoder_q.' + name[len('E.E.encoder_k.'):]].clone()
    return st
