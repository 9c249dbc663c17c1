"""nn.Module tree of the MI355X-native AirNet.  It keeps the reference's class names, constructor signature
`(opt)`, forward signatures and state_dict keys -- torch.nn layers are used only as
PARAMETER HOLDERS (default initialisation identical to the reference); every forward runs the HIP kernels
through fwair.functional.  Citations are file:line of the reference.

Supported (= every configuration of the reference that runs):
  Uformer encoder (encoder_msa_type freq | origin, L in {2, 3}) + Uformer decoder with
  degradation_embedding_method in {all_3_bands, all_DC, all_2_bands} or none.
Anything else raises NotImplementedError instead of silently diverging.
"""
import math

import os

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .lib import call

WIN = 8
# Tokens from which LeFF.run takes the fused forward kernel (csrc/fw_leff.hip).  OFF by default: measured on MI355X the fused kernel
# LOSES to the unfused chain on every high-resolution stage (tools/leff_probe.py: 1050 vs 490 us at C = 112, 262 144 tokens; 901 us
# even with its four twin stores removed) -- the 1.5x halo of GELU evaluations and the per-chunk weight fragments from L2 cost more
# than the hidden tensor's round trips through 6 TB/s of HBM save, and the unfused backward needs the twins written anyway.
_LEFF_FUSED_MIN_ROWS = int(os.environ.get('FW_LEFF_FUSED_MIN_ROWS', str(1 << 62)))


def trunc_normal_(t, std=.02):
    return nn.init.trunc_normal_(t, std=std)


def _tokens_hw(rows, batch):
    hw = rows // batch
    h = int(math.isqrt(hw))
    assert h * h == hw
    return h


# ---------------------------------------------------------------------------------------------------------------
# shared pieces
# ---------------------------------------------------------------------------------------------------------------
class LinearProjection(nn.Module):
    """decoder_Uformer.py:80-125 / encoder_Uformer.py:79-100 (plain self-attention projections)."""

    def __init__(self, dim, heads=8, dim_head=64, bias=True):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        self.to_q = nn.Linear(dim, inner, bias=bias)
        self.to_kv = nn.Linear(dim, inner * 2, bias=bias)

    def forward(self, xn):
        return Fn.QKVFn.apply(xn, self.to_q.weight, self.to_q.bias, self.to_kv.weight, self.to_kv.bias, getattr(self, '_fw_fused', None))


class LeFF(nn.Module):
    """net/utils/leff.py:71-117."""

    def __init__(self, dim=32, hidden_dim=128):
        super().__init__()
        self.linear1 = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU())
        self.conv = nn.Sequential(nn.Conv2d(hidden_dim, hidden_dim, groups=hidden_dim, kernel_size=3, stride=1, padding=1), nn.GELU())
        self.linear2 = nn.Sequential(nn.Linear(hidden_dim, dim))

    def run(self, xn, residual, rowscale, batch):
        rows, C = xn.shape
        h = _tokens_hw(rows, batch)
        fuse = None
        if (xn.dtype == torch.bfloat16 and C in (28, 56, 112) and h % 16 == 0 and rows >= _LEFF_FUSED_MIN_ROWS
                and self.linear1[0].weight.shape[0] == 4 * C):
            # high-resolution stages: the whole feed-forward in one kernel (csrc/fw_leff.hip); the three autograd nodes below
            # keep their backward passes, only their forward launches collapse into the first one
            fuse = dict(geo=(batch, h, h), residual=residual, rowscale=rowscale, rows_per_scale=h * h, wd=self.conv[0].weight,
                        bd=self.conv[0].bias, w2=self.linear2[0].weight, b2=self.linear2[0].bias)
        if fuse is not None:
            h1, g1 = Fn.linear(xn, self.linear1[0].weight, self.linear1[0].bias, gelu_out=True, fuse=fuse)
        else:                                      # g1 = GELU(h1) is never materialised: the depthwise kernels apply it on load
            h1, g1 = Fn.linear(xn, self.linear1[0].weight, self.linear1[0].bias), None
        h2, g2 = Fn.DwConvFn.apply(h1, g1, self.conv[0].weight, self.conv[0].bias, batch, h, h, fuse)
        return Fn.linear(g2, self.linear2[0].weight, self.linear2[0].bias, residual=residual, rowscale=rowscale,
                         rows_per_scale=h * h, x_pre=h2, fuse=fuse)

    def forward(self, x):                       # API parity: [B, HW, C] f32 -> [B, HW, C]
        B, HW, C = x.shape
        xn = Fn.CastFn.apply(x.reshape(B * HW, C))
        h1 = Fn.linear(xn, self.linear1[0].weight, self.linear1[0].bias)
        h2, g2 = Fn.DwConvFn.apply(h1, None, self.conv[0].weight, self.conv[0].bias, B, int(math.isqrt(HW)), int(math.isqrt(HW)))
        y = Fn.linear(g2, self.linear2[0].weight, self.linear2[0].bias, x_pre=h2, out_f32=True)
        return y.view(B, HW, C)


def _rel_index():
    ch = torch.arange(WIN)
    coords = torch.stack(torch.meshgrid([ch, ch], indexing='ij')).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += WIN - 1
    rel[:, :, 1] += WIN - 1
    rel[:, :, 0] *= 2 * WIN - 1
    return rel.sum(-1)


class Downsample(nn.Module):
    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=4, stride=2, padding=1))

    def run(self, x, batch):
        h = _tokens_hw(x.shape[0], batch)
        return Fn.DownsampleFn.apply(x, self.conv[0].weight, self.conv[0].bias, batch, h, h)


class Upsample(nn.Module):
    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.deconv = nn.Sequential(nn.ConvTranspose2d(in_channel, out_channel, kernel_size=2, stride=2))

    def run_cat(self, x, skip, batch):
        h = _tokens_hw(x.shape[0], batch)
        return Fn.UpsampleCatFn.apply(x, skip, self.deconv[0].weight, self.deconv[0].bias, batch, h, h)


class InputProj(nn.Module):
    def __init__(self, in_channel=3, out_channel=64):
        super().__init__()
        self.proj = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(inplace=True))

    def run(self, img):
        return Fn.InputProjFn.apply(img, self.proj[0].weight, self.proj[0].bias)


class OutputProj(nn.Module):
    def __init__(self, in_channel=64, out_channel=3):
        super().__init__()
        self.proj = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=1, padding=1))

    def run(self, fea, img):
        return Fn.OutputProjFn.apply(fea, self.proj[0].weight, self.proj[0].bias, img)


# ---------------------------------------------------------------------------------------------------------------
# decoder
# ---------------------------------------------------------------------------------------------------------------
def lfs_num_bands(methods):
    """decoder_Uformer.py:166-174 -> number of bands (0 = no frequency selection)."""
    nb = 0
    for t in methods:
        if 'all' not in t:
            continue
        if t.split('_')[-1] == 'bands':
            nb = int(t.split('_')[-2])
        elif t.split('_')[-1] == 'DC':
            nb = 2
    return nb


class DecWindowAttention(nn.Module):
    """decoder_Uformer.py:128-302 (`WindowAttention`)."""

    def __init__(self, input_resolution, dim, win_size, num_heads, all_degradation_embedding_method=()):
        super().__init__()
        self.input_resolution = input_resolution
        self.num_win = input_resolution[0] // win_size[0] * input_resolution[1] // win_size[1]
        self.dim, self.win_size, self.num_heads = dim, win_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.all_degradation_embedding_method = list(all_degradation_embedding_method)
        self.num_bands = lfs_num_bands(self.all_degradation_embedding_method)
        if self.num_bands > 0:
            if self.num_bands > 3:
                raise NotImplementedError('the HIP band filter covers all_3_bands / all_2_bands / all_DC')
            enc = 28 * 16                                                       # decoder_Uformer.py:176 hard-codes 28
            self.mlp_head = nn.ModuleList([nn.Sequential(nn.LayerNorm(enc), nn.Linear(enc, num_heads)) if i > 0 else None
                                           for i in range(self.num_bands)])
            self.avg = nn.ModuleList([nn.AdaptiveAvgPool1d(1) if i > 0 else None for i in range(self.num_bands)])
            self.mlp = nn.ModuleList([nn.Sequential(nn.Linear(num_heads, num_heads), nn.LeakyReLU(0.1, True),
                                                    nn.Linear(num_heads, num_heads)) if i > 0 else None
                                      for i in range(self.num_bands)])
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * win_size[0] - 1) * (2 * win_size[1] - 1), num_heads))
        self.register_buffer('relative_position_index', _rel_index())
        trunc_normal_(self.relative_position_bias_table)
        self.qkv = LinearProjection(dim, num_heads, dim // num_heads)
        self.proj = nn.Linear(dim, dim)

    def lambda_params(self):
        out = []
        for i in range(1, self.num_bands):
            out += [self.mlp_head[i][0].weight, self.mlp_head[i][0].bias, self.mlp_head[i][1].weight, self.mlp_head[i][1].bias,
                    self.mlp[i][0].weight, self.mlp[i][0].bias, self.mlp[i][2].weight, self.mlp[i][2].bias]
        return out

    @property
    def lfs_mode(self):
        return 0 if self.num_bands == 0 else (2 if self.num_bands == 3 else 1)


class DecLeWinTransformerBlock(nn.Module):
    """decoder_Uformer.py:504-756 (`LeWinTransformerBlock`, plain + all_* path)."""

    def __init__(self, dim, input_resolution, num_heads, win_size=8, shift_size=0, mlp_ratio=4., drop_path=0.,
                 all_degradation_embedding_method=(), debug_mode=False):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.win_size, self.shift_size = win_size, shift_size
        if min(input_resolution) <= win_size:                                   # :531-533
            self.shift_size = 0
            self.win_size = min(input_resolution)
        if self.win_size != WIN:
            raise NotImplementedError('feature maps smaller than one 8x8 window')
        self.debug_mode = bool(debug_mode)
        if self.debug_mode:                                                     # :518-519 (no parameters, no buffers: same state_dict)
            from net.utils.frequency_decompose import FrequencyDecompose
            self.visual_decompose = FrequencyDecompose('frequency_decompose', 1, input_resolution[0], input_resolution[1], inverse='visual')
        self.norm1 = nn.LayerNorm(dim)
        self.attn = DecWindowAttention(input_resolution, dim, (self.win_size, self.win_size), num_heads,
                                       all_degradation_embedding_method)
        self.drop_path_rate = float(drop_path)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = LeFF(dim, int(dim * mlp_ratio))
        self._name = ''

    def _spectrum(self, t, batch, h):
        """decoder_Uformer.py:668-673 / :731-736: magnitude spectrum (HIP DFT, `visual` mode) of a token-major activation as a
        [B, C, H, W] map, averaged over batch and channels -> [H, W]."""
        img = t.detach().float().reshape(batch, h, h, t.shape[1]).permute(0, 3, 1, 2).contiguous()
        return self.visual_decompose(img).squeeze(0).mean(0).mean(0)

    def run(self, x, batch, coef, visual=None):
        """visual: list that receives this block's debug payload [spectrum_before, spectrum_after, embed_lamb] (:753-754)."""
        rows, C = x.shape
        h = _tokens_hw(rows, batch)
        # two independent masks per block, as the reference's two self.drop_path(...) calls draw (decoder_Uformer.py:739,751)
        dp = Fn.droppath_scale(self._name + 'attn', batch, self.drop_path_rate, self.training, x.device)
        dp2 = Fn.droppath_scale(self._name + 'mlp', batch, self.drop_path_rate, self.training, x.device)
        x, xn = Fn.LnResFn.apply(x, self.norm1.weight, self.norm1.bias)
        qkv = self.attn.qkv(xn)
        geo = (C, batch, h, h, self.num_heads, 1, 0, self.shift_size, self.attn.lfs_mode if coef is not None else 0)
        tab = self.attn.relative_position_bias_table
        o = Fn.WindowAttnFn.apply(qkv, tab.unsqueeze(0), coef, geo, tab, getattr(coef, '_fw_dgrad', None))
        if visual is not None:
            # the attention branch is needed on its own (before the residual add): plain projection, then the add
            y = Fn.linear(o, self.attn.proj.weight, self.attn.proj.bias, out_f32=True)
            lamb = []
            if coef is not None:           # (a, b, c) = (1 + lambda_last, -lambda_last / 64, ...): the LAST band's lambda, [B, 1, heads] (:277-296)
                lamb = (coef.detach()[:, :, 0] - 1.0).reshape(batch, 1, self.num_heads)
            visual.append([self._spectrum(xn, batch, h), self._spectrum(y, batch, h), lamb])
            x = x + (y if dp is None else y * dp.repeat_interleave(h * h)[:, None])
        else:
            x = Fn.linear(o, self.attn.proj.weight, self.attn.proj.bias, residual=x, rowscale=dp, rows_per_scale=h * h)
        x, xn2 = Fn.LnResFn.apply(x, self.norm2.weight, self.norm2.bias)
        return self.mlp.run(xn2, x, dp2, batch)


class DecBasicUformerLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, win_size, mlp_ratio, drop_path, methods, debug_mode=False):
        super().__init__()
        self.blocks = nn.ModuleList([
            DecLeWinTransformerBlock(dim, input_resolution, num_heads, win_size, 0 if i % 2 == 0 else win_size // 2, mlp_ratio,
                                     drop_path[i] if isinstance(drop_path, list) else drop_path, methods, debug_mode)
            for i in range(depth)])


def _resolve_img_size(opt, img_size):
    """The reference's seam calls `cls(opt)` and so always builds for img_size=128 (net/model.py:17,31), whatever `--patch_size`
    says; a 64- or 256-pixel patch then fails inside its window partition.  Here the constructor default follows
    `opt.patch_size`, so the same seam builds the 256x256 model the reference classes give with img_size=256.
    The bottleneck (S/16) must hold whole 8x8 windows: S is a multiple of 128."""
    S = int(img_size if img_size is not None else (getattr(opt, 'patch_size', None) or 128))
    if S < 128 or S % 128:
        raise NotImplementedError(f'img_size={S}: the five-stage Uformer with 8x8 windows needs a multiple of 128 '
                                  '(the reference fails in window_partition otherwise)')
    return S


class UformerDecoder(nn.Module):
    """decoder_Uformer.py:835-1171."""

    def __init__(self, opt, img_size=None, in_chans=3, out_chans=3, depths=(2, 2, 8, 8, 2, 8, 8, 2, 2),
                 num_heads=(1, 2, 4, 8, 16, 16, 8, 4, 2), win_size=8, mlp_ratio=4., drop_path_rate=0.1):
        super().__init__()
        self.opt = opt
        img_size = _resolve_img_size(opt, img_size)
        self.debug_mode = bool(getattr(opt, 'debug_mode', False))               # :847 -- forward then returns (restored, visual_freqs)
        for m in opt.degradation_embedding_method:
            if 'all' not in m and m not in ('None', 'none'):
                raise NotImplementedError(f'degradation_embedding_method={m!r} does not run in the reference either '
                                          '(SURVEY.md 0.1); supported: all_3_bands, all_2_bands, all_DC')
        E = opt.embed_dim
        self.embed_dim, self.win_size, self.reso, self.in_chans = E, win_size, img_size, in_chans
        methods = [t for t in opt.degradation_embedding_method if 'all' in t]
        depths = list(depths)
        enc_dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths[:4]))]
        conv_dpr = [drop_path_rate] * depths[4]
        dec_dpr = enc_dpr[::-1]
        self.input_proj = InputProj(in_chans, E)
        self.output_proj = OutputProj(2 * E, out_chans)
        mk = lambda dim, res, d, h, dpr: DecBasicUformerLayer(dim, (res, res), d, h, win_size, mlp_ratio, dpr, methods, self.debug_mode)
        o = 0
        for i in range(4):
            setattr(self, f'encoderlayer_{i}', mk(E * 2 ** i, img_size // 2 ** i, depths[i], num_heads[i], enc_dpr[o:o + depths[i]]))
            setattr(self, f'dowsample_{i}', Downsample(E * 2 ** i, E * 2 ** (i + 1)))
            o += depths[i]
        self.bottleneck_0 = mk(E * 16, img_size // 16, depths[4], num_heads[4], conv_dpr)
        self.bottleneck_1 = mk(E * 16, img_size // 16, depths[4], num_heads[4], conv_dpr)
        self.upsample_3 = Upsample(E * 16, E * 8)
        self.decoderlayer_3 = mk(E * 16, img_size // 8, depths[5], num_heads[5], dec_dpr[:depths[5]])
        self.upsample_2 = Upsample(E * 16, E * 4)
        self.decoderlayer_2 = mk(E * 8, img_size // 4, depths[6], num_heads[6], dec_dpr[sum(depths[5:6]):sum(depths[5:7])])
        self.upsample_1 = Upsample(E * 8, E * 2)
        self.decoderlayer_1 = mk(E * 4, img_size // 2, depths[7], num_heads[7], dec_dpr[sum(depths[5:7]):sum(depths[5:8])])
        self.upsample_0 = Upsample(E * 4, E)
        self.decoderlayer_0 = mk(E * 2, img_size, depths[8], num_heads[8], dec_dpr[sum(depths[5:8]):sum(depths[5:9])])
        self.apply(self._init_weights)                                          # :1094-1104
        self._order = ['encoderlayer_0', 'encoderlayer_1', 'encoderlayer_2', 'encoderlayer_3', 'bottleneck_0', 'bottleneck_1',
                       'decoderlayer_3', 'decoderlayer_2', 'decoderlayer_1', 'decoderlayer_0']
        for ln in self._order:
            for bi, blk in enumerate(getattr(self, ln).blocks):
                blk._name = f'R.R.{ln}.blocks.{bi}.'

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def all_blocks(self):
        return [blk for ln in self._order for blk in getattr(self, ln).blocks]

    def _coefs(self, inter, B):
        """(a, b, c) of every block from the encoder representation, one launch (decoder_Uformer.py:279-288)."""
        blocks = self.all_blocks()
        nb = blocks[0].attn.num_bands
        if nb == 0:
            return [None] * len(blocks)
        if isinstance(inter, (tuple, list)):
            stack = getattr(inter[0], '_fw_stack', None)                        # the encoder's own [L, B, (S/16)^2, C] buffer
            if stack is None:
                stack = torch.stack([t.float() for t in inter[:nb]], 0)
        else:
            stack = inter
        C = stack.shape[-1]
        NT = stack.shape[-2]
        assert C == 448, 'LFS heads expect the [B, (S/16)^2, 448] encoder representation'
        bands = stack[1:nb].reshape((nb - 1) * B * NT, C)
        params = [p for blk in blocks for p in blk.attn.lambda_params()]
        heads = tuple(blk.num_heads for blk in blocks)
        coef = Fn.LfsLambdaFn.apply(bands, (heads, B, nb - 1), *params)
        out, o = [], 0
        dgrad = getattr(coef.grad_fn, 'dcoef', None) if coef.grad_fn is not None else None     # LfsLambdaFn's accumulation buffer
        for h in heads:
            c = coef[o:o + B * h * 3].view(B, h, 3)
            if dgrad is not None:
                c._fw_dgrad = dgrad[o:o + B * h * 3].view(B, h, 3)
            out.append(c)
            o += B * h * 3
        return out

    def forward(self, x, inter, mask=None):
        assert mask is None
        B = x.shape[0]
        if x.shape[-1] != self.reso or x.shape[-2] != self.reso:
            raise NotImplementedError(f'the decoder is built for {self.reso}x{self.reso} inputs (decoder_Uformer.py:836)')
        coefs = iter(self._coefs(inter, B))
        x = x.contiguous().float()
        y = self.input_proj.run(x)

        visual_freqs = [] if self.debug_mode else None                          # :1130, one list per layer in forward order

        def layer(name, t):
            vis = None
            if visual_freqs is not None:
                vis = []
                visual_freqs.append(vis)
            for blk in getattr(self, name).blocks:
                t = blk.run(t, B, next(coefs), vis)
            return t

        conv = []
        for i in range(4):
            y = layer(f'encoderlayer_{i}', y)
            conv.append(y)
            y = getattr(self, f'dowsample_{i}').run(y, B)
        y = layer('bottleneck_0', y)
        y = layer('bottleneck_1', y)
        for i in reversed(range(4)):
            y = getattr(self, f'upsample_{i}').run_cat(y, conv[i], B)
            y = layer(f'decoderlayer_{i}', y)
        out = self.output_proj.run(y, x)
        if self.debug_mode:                                                     # :1168-1169
            return out, visual_freqs
        return out


# ---------------------------------------------------------------------------------------------------------------
# encoder
# ---------------------------------------------------------------------------------------------------------------
class EncWindowAttention(nn.Module):
    """encoder_Uformer.py:103-186 (`WindowAttention`, encoder_msa_type == 'origin')."""

    def __init__(self, dim, win_size, num_heads):
        super().__init__()
        self.dim, self.win_size, self.num_heads = dim, win_size, num_heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * win_size[0] - 1) * (2 * win_size[1] - 1), num_heads))
        self.register_buffer('relative_position_index', _rel_index())
        trunc_normal_(self.relative_position_bias_table)
        self.qkv = LinearProjection(dim, num_heads, dim // num_heads)
        self.proj = nn.Linear(dim, dim)

    def tables(self):
        return self.relative_position_bias_table.unsqueeze(0)


class FrequencyWindowAttention(nn.Module):
    """encoder_Uformer.py:190-313."""

    def __init__(self, dim, win_size, num_heads, type=None, L=3):
        super().__init__()
        assert type in ('intra', 'inter'), 'Attention type error.'
        self.dim, self.win_size, self.num_heads, self.L, self.type = dim, win_size, num_heads, L, type
        self.relative_position_bias_table = nn.ParameterList([
            nn.Parameter(torch.zeros((2 * win_size[0] - 1) * (2 * win_size[1] - 1), num_heads)) for _ in range(L * L)])
        self.register_buffer('relative_position_index', _rel_index())
        for i in range(L * L):
            trunc_normal_(self.relative_position_bias_table[i])
        self.qkv = LinearProjection(dim, num_heads, dim // num_heads)
        self.proj = nn.Linear(dim, dim)
        n = win_size[0] * win_size[1]
        eye = torch.eye(L)
        mf = (1 - eye) * -100.0 if type == 'intra' else eye * -100.0          # :246-254
        self.register_buffer('mask_freq', mf.repeat_interleave(n, 0).repeat_interleave(n, 1)[None, None])

    def tables(self):
        t = getattr(self, '_fw_tab', None)                # engine: the tables lie packed in the flat buffer (TrainEngine._fuse_tables)
        if t is not None and t.data_ptr() == self.relative_position_bias_table[0].data_ptr():
            return t
        return torch.stack(list(self.relative_position_bias_table), 0)

    def table_grads(self):
        """Holder of the dense gradient view the backward kernel accumulates into (engine mode), else None (autograd path)."""
        t = getattr(self, '_fw_tab', None)
        if t is not None and t.data_ptr() == self.relative_position_bias_table[0].data_ptr():
            return getattr(self, '_fw_tab_grad', None)
        return None


class EncLeWinTransformerBlock(nn.Module):
    """encoder_Uformer.py:515-682."""

    def __init__(self, dim, input_resolution, num_heads, win_size=8, shift_size=0, mlp_ratio=4., drop_path=0.,
                 encoder_msa_type=None, L=3):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.win_size, self.shift_size, self.L = win_size, shift_size, L
        if min(input_resolution) <= win_size:
            self.shift_size = 0
            self.win_size = min(input_resolution)
        if self.win_size != WIN:
            raise NotImplementedError('feature maps smaller than one 8x8 window')
        self.norm1 = nn.LayerNorm(dim)
        self.encoder_msa_type = encoder_msa_type
        ws = (self.win_size, self.win_size)
        if encoder_msa_type == 'origin':
            self.attn = EncWindowAttention(dim, ws, num_heads)
        elif encoder_msa_type == 'freq':
            if L not in (2, 3):
                raise NotImplementedError('frequency attention is built for L in {2, 3}')
            self.attn_intra = FrequencyWindowAttention(dim, ws, num_heads, type='intra', L=L)
            self.attn_inter = FrequencyWindowAttention(dim, ws, num_heads, type='inter', L=L)
        else:
            raise AssertionError('MSA type error.')
        self.drop_path_rate = float(drop_path)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = LeFF(dim, int(dim * mlp_ratio))
        self._name = ''

    def run(self, x, nimg):
        """x: f32 [(l b) H W, C];  nimg = L*B images."""
        rows, C = x.shape
        h = _tokens_hw(rows, nimg)
        # two independent masks per block (encoder_Uformer.py:679-680)
        dp = Fn.droppath_scale(self._name + 'attn', nimg, self.drop_path_rate, self.training, x.device)
        dp2 = Fn.droppath_scale(self._name + 'mlp', nimg, self.drop_path_rate, self.training, x.device)
        x, xn = Fn.LnResFn.apply(x, self.norm1.weight, self.norm1.bias)
        if self.encoder_msa_type == 'origin':
            geo = (C, nimg, h, h, self.num_heads, 1, 0, self.shift_size, 0)
            o = Fn.WindowAttnFn.apply(self.attn.qkv(xn), self.attn.tables(), None, geo, self.attn.relative_position_bias_table, None)
            x = Fn.linear(o, self.attn.proj.weight, self.attn.proj.bias, residual=x, rowscale=dp, rows_per_scale=h * h)
        else:
            B = nimg // self.L
            a = self.attn_intra
            o = Fn.WindowAttnFn.apply(a.qkv(xn), a.tables(), None, (C, B, h, h, self.num_heads, self.L, 0, self.shift_size, 0), a.table_grads(), None)
            y1 = Fn.linear(o, a.proj.weight, a.proj.bias)
            a = self.attn_inter
            o = Fn.WindowAttnFn.apply(a.qkv(y1), a.tables(), None, (C, B, h, h, self.num_heads, self.L, 1, self.shift_size, 0), a.table_grads(), None)
            x = Fn.linear(o, a.proj.weight, a.proj.bias, residual=x, rowscale=dp, rows_per_scale=h * h)
        x, xn2 = Fn.LnResFn.apply(x, self.norm2.weight, self.norm2.bias)
        return self.mlp.run(xn2, x, dp2, nimg)


class EncBasicUformerLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, win_size, mlp_ratio, drop_path, msa, L):
        super().__init__()
        self.blocks = nn.ModuleList([
            EncLeWinTransformerBlock(dim, input_resolution, num_heads, win_size, 0 if i % 2 == 0 else win_size // 2, mlp_ratio,
                                     drop_path[i] if isinstance(drop_path, list) else drop_path, msa, L)
            for i in range(depth)])


class Uformer(nn.Module):
    """Encoder body, encoder_Uformer.py:746-923."""

    def __init__(self, opt, img_size=128, in_chans=3, depths=(2, 2, 2, 2, 2, 2, 2, 2, 2), num_heads=(1, 2, 4, 8, 16, 16, 8, 4, 2),
                 win_size=8, mlp_ratio=4., drop_path_rate=0.1, **kw):
        super().__init__()
        self.opt = opt
        E, L, msa = opt.encoder_embed_dim, opt.L, opt.encoder_msa_type
        depths = list(depths)
        enc_dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths[:4]))]
        conv_dpr = [drop_path_rate] * depths[4]
        self.input_proj = InputProj(in_chans, E)
        o = 0
        for i in range(4):
            setattr(self, f'encoderlayer_{i}', EncBasicUformerLayer(E * 2 ** i, (img_size // 2 ** i,) * 2, depths[i], num_heads[i],
                                                                    win_size, mlp_ratio, enc_dpr[o:o + depths[i]], msa, L))
            setattr(self, f'dowsample_{i}', Downsample(E * 2 ** i, E * 2 ** (i + 1)))
            o += depths[i]
        self.conv = EncBasicUformerLayer(E * 16, (img_size // 16,) * 2, depths[4], num_heads[4], win_size, mlp_ratio, conv_dpr, msa, L)
        self.apply(UformerDecoder._init_weights)

    def run(self, img, nimg):
        y = self.input_proj.run(img)
        for i in range(4):
            for blk in getattr(self, f'encoderlayer_{i}').blocks:
                y = blk.run(y, nimg)
            y = getattr(self, f'dowsample_{i}').run(y, nimg)
        for blk in self.conv.blocks:
            y = blk.run(y, nimg)
        return y


class UformerEncoder(nn.Module):
    """encoder_Uformer.py:926-986."""

    def __init__(self, opt, img_size=None, in_chans=3, out_chans=3):
        super().__init__()
        from net.utils.frequency_decompose import FrequencyDecompose
        self.opt = opt
        img_size = _resolve_img_size(opt, img_size)
        E = opt.encoder_embed_dim
        self.img_size = img_size
        if not opt.L == 1:
            self.preprocess_decompose = FrequencyDecompose('frequency_decompose_1', 1. / (opt.L - 1), img_size, img_size)
        self.uformer = Uformer(opt, img_size=img_size, in_chans=in_chans)
        self.mlp_head = nn.ModuleList([nn.Sequential(nn.LayerNorm(E * 16), nn.Linear(E * 16, opt.encoder_dim * 16 * 16))
                                       for _ in range(opt.L)])
        self.norm = nn.ModuleList([nn.Sequential(nn.BatchNorm2d(opt.encoder_dim), nn.LeakyReLU(0.1, True)) for _ in range(opt.L)])
        self.avg = nn.ModuleList([nn.AdaptiveAvgPool2d(1) for _ in range(opt.L)])
        self.mlp = nn.ModuleList([nn.Sequential(nn.Linear(opt.encoder_dim, opt.encoder_dim), nn.LeakyReLU(0.1, True),
                                                nn.Linear(opt.encoder_dim, opt.encoder_dim)) for _ in range(opt.L)])
        self.set_prefix('')

    def set_prefix(self, prefix):
        for name, blk in self.uformer.named_modules():
            if isinstance(blk, EncLeWinTransformerBlock):
                blk._name = f'{prefix}uformer.{name}.'

    def forward(self, x, mask=None, want_heads=True):
        assert mask is None
        opt, L = self.opt, self.opt.L
        B = x.shape[0]
        if x.shape[-1] != self.img_size or x.shape[-2] != self.img_size:
            raise NotImplementedError(f'the encoder is built for {self.img_size}x{self.img_size} inputs (encoder_Uformer.py:927)')
        x = x.contiguous().float()
        if L != 1:
            x = self.preprocess_decompose(x).reshape(L * B, *x.shape[1:])      # 'l b c h w -> (l b) c h w'
        y = self.uformer.run(x, L * B)                                          # f32 [(l b) 64, 448]
        C = y.shape[1]
        stack = y.view(L, B, y.shape[0] // (L * B), C)
        inter = tuple(stack.unbind(0))
        inter[0]._fw_stack = stack
        out = []
        if want_heads:
            for i in range(L):
                xi = stack[i].reshape(-1, C)
                xn = Fn.LayerNormFn.apply(xi, self.mlp_head[i][0].weight, self.mlp_head[i][0].bias)
                fea = Fn.linear(xn, self.mlp_head[i][1].weight, self.mlp_head[i][1].bias)
                bn = self.norm[i][0]
                gap = Fn.BnLreluGapFn.apply(fea.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                            bn.num_batches_tracked, B, self.training)
                g = Fn.CastFn.apply(gap)
                hmid = Fn.linear(g, self.mlp[i][0].weight, self.mlp[i][0].bias, out_f32=True)
                out.append(Fn.linear(Fn.LreluFn.apply(hmid, 0.1), self.mlp[i][2].weight, self.mlp[i][2].bias, out_f32=True))
        return None, out, inter


# ---------------------------------------------------------------------------------------------------------------
# MoCo + AirNet  (net/utils/moco.py, net/model.py)
# ---------------------------------------------------------------------------------------------------------------
_KEY_STREAMS = {}


class MoCo(nn.Module):
    """net/utils/moco.py:6-170.  The dead DDP helpers (:68-113,174-185) are not reproduced."""

    def __init__(self, opt, base_encoder, dim, K=3 * 256, m=0.999, T=0.07, mlp=False):
        super().__init__()
        self.num_losses = opt.L
        self.opt, self.K, self.m, self.T = opt, K, m, T
        self.encoder_q = base_encoder(opt)
        self.encoder_k = base_encoder(opt)
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            pk.data.copy_(pq.data)
            pk.requires_grad = False
        self.register_buffer('queue', torch.randn(self.num_losses, dim, K))
        for i in range(self.num_losses):
            self.queue[i] = nn.functional.normalize(self.queue[i], dim=0)
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        if hasattr(self.encoder_q, 'set_prefix'):
            self.encoder_q.set_prefix('E.E.encoder_q.')
            self.encoder_k.set_prefix('E.E.encoder_k.')
        self._ema_hook = None

    def _key_stream(self, device):
        if os.environ.get('FW_KEY_STREAM', '1') == '0' or device.type != 'cuda':
            return None
        side = _KEY_STREAMS.get(device)                  # module-level: a stream attribute would break copy.deepcopy(net)
        if side is None:
            if torch.cuda.is_current_stream_capturing():
                return None                              # never create a stream inside a capture; the warm-up steps did already
            side = _KEY_STREAMS[device] = torch.cuda.Stream(device=device)
        return side

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        if self._ema_hook is not None:                   # engine: one launch over the flat parameter buffers
            self._ema_hook()
            return
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            call('fw_ema', 0, pk.data, pq.data, None, pk.numel(), self.m)
        Fn.config.shadow_epoch += 1                      # raw-pointer updates do not bump tensor versions

    @torch.no_grad()
    def _dequeue_and_enqueue(self, khat):
        L, B, ED = khat.shape
        assert self.K % B == 0                           # moco.py:59
        call('fw_moco_enqueue', self.queue, khat, self.queue_ptr, L, B, ED, self.K)

    def forward(self, im_q, im_k):
        if not self.training:
            embedding, _, inter = self.encoder_q(im_q, want_heads=False)
            return embedding, inter
        return self.forward_finish(self.forward_begin(im_q, im_k))

    def forward_begin(self, im_q, im_k):
        """Train-mode forward up to the encoders' outputs (moco.py:115-141).  The key branch (EMA update + key-encoder forward)
        depends on nothing the query forward produces, and the deep stages of either encoder launch fewer workgroups than the
        chip has CUs: it runs on a second HIP stream, forked here and joined in `forward_finish` (inside a captured step this
        is a fork / join in the graph: 217.6 -> 221.0 images/s).  Joining later, after the decoder forward, measured the same.  """
        side = self._key_stream(im_q.device)
        k = None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                self._momentum_update_key_encoder()
                _, k, _ = self.encoder_k(im_k)
        embedding, q, inter = self.encoder_q(im_q)
        if side is None:
            with torch.no_grad():
                self._momentum_update_key_encoder()
                _, k, _ = self.encoder_k(im_k)
        return embedding, q, inter, k, side

    def forward_finish(self, state):
        embedding, q, inter, k, side = state
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        n = len(q)                                       # = L for the Uformer encoder (moco.py:127 indexes range(L))
        qs, ks = torch.stack(q, 0), torch.stack(k, 0)
        logits, khat = Fn.MocoLogitsFn.apply(qs, ks, self.queue[:n], self.T)
        labels = [torch.zeros(logits.shape[1], dtype=torch.long, device=logits.device) for _ in range(n)]
        self._dequeue_and_enqueue(khat)
        return embedding, list(logits.unbind(0)), labels, inter
