"""The convolutional plug-ins of the model seam on the HIP kernels of csrc/fw_conv.hip (BASELINE configs[0]):
`ResNetEncoder` (net/encoder_ResNet.py:4-47) and `DGRN` = `ResNetDecoder` (net/decoder_DGRN.py:9-158, net/model.py:3) with the
modulated deformable convolution of net/utils/deform_conv.py:10-67.  Same class names, constructor signature `(opt)`, forward
signatures and state_dict keys as the reference (tests/golden/schema.json `resnet_dgrn`, dumped from it); torch.nn layers are
parameter holders only, every forward / backward body is a sequence of C-ABI calls.

Activations are token-major T tensors [B*H*W, C] (T = fwair.functional.config.compute_dtype).  At the API boundary the reference's
NCHW shapes are kept: `inter` is handed out as an NCHW *view* of the token-major buffer (no copy), images enter and leave as f32
[B, 3, H, W].

DCN_layer: the reference's forward ends in `assert False` (deform_conv.py:64; the mmcv op is commented out), so this arithmetic has
no runnable reference anywhere -- it follows the published DCNv2 definition (see oracle/convnets_oracle.py) and is checked against
that restatement and by known-answer tests: parity unpinned.
"""
import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .lib import call, dt


def _chunk(dtype):
    return 32 if dtype == torch.bfloat16 else 16           # elements per 64-byte MFMA K chunk


def _rup(v, m):
    return (v + m - 1) // m * m


def _zeros(shape, dev, dtype=torch.float32):
    return torch.zeros(shape, dtype=dtype, device=dev)


# ---------------------------------------------------------------------------------------------------------------
# weight panels: T [roundup(Cout, 16)][9 * Cinp], element (co, tap, ci), cached per parameter version (functional.shadow's stamps)
# ---------------------------------------------------------------------------------------------------------------
_panels = {}


def panel(param, kind):
    """kind 'fwd': [Cop][9 * Cip] (a [Co, Ci, 1, 1] weight sits at the centre tap);  'dgrad': the flipped, transposed panel
    [roundup(Ci, 16)][9 * Coc] with element (ci, 8 - tap, co) -- the input gradient of a stride-1 convolution is a convolution."""
    dtype = Fn.config.compute_dtype
    p = param.detach()
    key = (id(param), kind, dtype)
    stamp = (param._version, Fn.config.shadow_epoch, p.data_ptr())
    hit = _panels.get(key)
    if hit is not None and hit[0] == stamp and hit[2]() is param:
        return hit[1]
    Co, Ci, kh, _ = p.shape
    ck = _chunk(dtype)
    src = p.contiguous().reshape(Co, Ci, kh * kh)
    if kind == 'fwd':
        Cip = _rup(Ci, ck)
        out = _zeros((_rup(Co, 16), 9 * Cip), p.device, dtype)
        off = 0 if kh == 3 else 4 * Cip
        ops.permute3(src, out.view(-1)[off:], (Co, Ci, kh * kh), (9 * Cip, 1, Cip if kh == 3 else 0))
    else:
        assert kh == 3
        Coc = _rup(Co, ck)
        out = _zeros((_rup(Ci, 16), 9 * Coc), p.device, dtype)
        ops.permute3(src, out.view(-1)[8 * Coc:], (Co, Ci, 9), (1, 9 * Coc, -Coc))
    import weakref
    _panels[key] = (stamp, out, weakref.ref(param))
    return out


def tokens(img, Cp=None):
    """f32 [B, Ci, H, W] -> T [B*H*W, Cp] token-major, channels zero-padded to one MFMA chunk."""
    B, Ci, H, W = img.shape
    dtype = Fn.config.compute_dtype
    Cp = Cp or _rup(Ci, _chunk(dtype))
    out = torch.empty((B * H * W, Cp), dtype=dtype, device=img.device)
    call('fw_nchw_to_tokens', dt(dtype), img.contiguous().float(), out, Cp, B, Ci, H * W, Cp)
    return out


class TokensFn(torch.autograd.Function):
    """image planes -> padded token map (differentiable: the decoder's input image needs no gradient, a feature map might)."""

    @staticmethod
    def forward(ctx, img):
        ctx.shape = img.shape
        return tokens(img)

    @staticmethod
    def backward(ctx, dy):
        B, Ci, H, W = ctx.shape
        out = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        call('fw_tokens_to_nchw', dt(dy.dtype), dy, dy.stride(0), out, B, Ci, H * W)
        return out


class ToImageFn(torch.autograd.Function):
    """token map T [B*H*W, ld] (first Ci channels) -> f32 [B, Ci, H, W]."""

    @staticmethod
    def forward(ctx, tok, B, Ci, H, W):
        ctx.geo = (B, Ci, H, W, tok.shape[1], tok.dtype)
        out = torch.empty((B, Ci, H, W), dtype=torch.float32, device=tok.device)
        call('fw_tokens_to_nchw', dt(tok.dtype), tok, tok.stride(0), out, B, Ci, H * W)
        return out

    @staticmethod
    def backward(ctx, dimg):
        B, Ci, H, W, Cp, dtype = ctx.geo
        d = torch.empty((B * H * W, Cp), dtype=dtype, device=dimg.device)
        call('fw_nchw_to_tokens', dt(dtype), dimg.contiguous().float(), d, Cp, B, Ci, H * W, Cp)
        return d, None, None, None, None


# ---------------------------------------------------------------------------------------------------------------
# convolution  (3x3 p1 or 1x1 p0, stride 1 | 2) on token maps
# ---------------------------------------------------------------------------------------------------------------
class ConvFn(torch.autograd.Function):
    """y = act(conv(cat[x, x2]) + bias) + res.   x: T [B*H*W, >= Ci1];  x2: optional second channel group (DCN's cat[x, inter]);
    weight: parameter [Co, Ci, k, k], k in {1, 3};  act: LeakyReLU slope or None;  out_f32: y is f32 [Mo, roundup(Co, 8)]."""

    @staticmethod
    def forward(ctx, x, x2, weight, bias, res, geo, stride, slope, out_f32):
        B, H, W = geo
        Co, Ci, k, _ = weight.shape
        dtype = x.dtype
        ck = _chunk(dtype)
        Cip = _rup(Ci, ck)
        Ci1 = Cip if x2 is None else x.shape[1]
        assert x.shape[1] >= Ci1 and x.shape[0] == B * H * W and (x2 is None or Ci1 + x2.shape[1] == Ci)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        ldo = _rup(Co, 8)
        y = (torch.zeros if ldo != Co else torch.empty)((B * Ho * Wo, ldo), dtype=torch.float32 if out_f32 else dtype, device=x.device)
        w = panel(weight, 'fwd')
        call('fw_conv3x3', dt(dtype), x, x.stride(0), x2, x2.stride(0) if x2 is not None else 0, Cip, Ci1, w, bias, y, ldo, int(out_f32),
             res, res.stride(0) if res is not None else 0, Co, B, H, W, stride, 0x1ff if k == 3 else 0x010, 1 if slope is not None else 0,
             float(slope or 0.0))
        ctx.save_for_backward(x, x2, weight, y if slope is not None else None)
        ctx.cfg = (geo, stride, slope, out_f32, bias is not None, res is not None, Ci1, Cip)
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x2, weight, y = ctx.saved_tensors
        (B, H, W), stride, slope, out_f32, has_bias, has_res, Ci1, Cip = ctx.cfg
        Co, Ci, k, _ = weight.shape
        dtype = x.dtype
        ck = _chunk(dtype)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        Mo = B * Ho * Wo
        dres = dy if has_res else None
        # ---- gradient at the pre-activation as a T map whose channels are padded to whole chunks (the dgrad convolution's input)
        Coc = _rup(Co, ck)
        if slope is not None:
            g = (torch.empty if Coc == Co else torch.zeros)((Mo, Coc), dtype=dtype, device=dy.device)
            call('fw_lrelu_t_bwd', dt(dtype), dy, dy.stride(0), y, y.stride(0), g, Coc, Mo, Co, float(slope))
        elif dy.dtype != dtype:                                  # f32 output (offset / mask logits, [Mo, 32]): cast, padding included
            g = torch.empty((Mo, Coc), dtype=dtype, device=dy.device)
            call('fw_cast_rows', dt(dtype), dy.contiguous(), dy.shape[1], g, Coc, Mo, Coc, None, 1)
        elif Coc == Co and dy.stride(1) == 1 and (dy.stride(0) * dy.element_size()) % 16 == 0 and dy.data_ptr() % 16 == 0:
            g = dy
        else:
            g = _zeros((Mo, Coc), dy.device, dtype)
            g[:, :Co].copy_(dy[:, :Co])
        # ---- weight / bias gradient: explicit [Mo][9 * Cip] operand + the split-K GEMM of fw_gemm (dW = g^T col, db = column sums of g)
        xin = x if x2 is None else torch.cat([x[:, :Ci1], x2], 1)
        col = torch.empty((Mo, 9 * Cip), dtype=dtype, device=x.device)
        call('fw_im2col3', dt(dtype), xin, xin.stride(0), col, B, H, W, Cip, stride)
        dwk = _zeros((Co, 9 * Cip), x.device)
        db = _zeros((Co,), x.device) if has_bias else None
        ops.wgrad(g[:, :Co], col, Co, 9 * Cip, Mo, dwk, db)
        dw = dwk.view(Co, 9, Cip)[:, :, :Ci]
        dw = (dw.permute(0, 2, 1).reshape(Co, Ci, 3, 3) if k == 3 else dw[:, 4].reshape(Co, Ci, 1, 1)).contiguous()
        del col
        # ---- input gradient: a stride-1 3x3 convolution's is again an implicit-GEMM convolution (flipped, transposed panel);
        # stride 2 and 1x1: d(col) = g W by GEMM, then the gather form of col2im
        dx = dx2 = None
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]):
            if stride == 1 and k == 3:
                d = torch.empty((Mo, _rup(Ci, 8)), dtype=dtype, device=x.device)
                call('fw_conv3x3', dt(dtype), g, Coc, None, 0, Coc, Coc, panel(weight, 'dgrad'), None, d, d.stride(0), 0, None, 0, Ci, B, H, W,
                     1, 0x1ff, 0, 0.0)
            else:
                dcol = ops.gemm(g[:, :Co], panel(weight, 'fwd'), Mo, 9 * Cip, Co, w_trans=True)
                d = torch.empty((B * H * W, Cip), dtype=dtype, device=x.device)
                call('fw_col2im3', dt(dtype), dcol, d, Cip, B, H, W, Cip, stride)
            if x2 is None:
                dx = _fit(d, x.shape[1])
            else:
                dx, dx2 = d[:, :Ci1], d[:, Ci1:Ci]
        return dx, dx2, dw, db, dres, None, None, None, None


def _fit(d, cols):
    """Gradient of a channel-padded token map: same column count as the forward input (extra columns are padding)."""
    if d.shape[1] == cols:
        return d
    if d.shape[1] > cols:
        return d[:, :cols]
    out = _zeros((d.shape[0], cols), d.device, d.dtype)
    out[:, :d.shape[1]] = d
    return out


def conv(x, mod, geo, stride=1, slope=None, res=None, x2=None, out_f32=False):
    return ConvFn.apply(x, x2, mod.weight, mod.bias, res, geo, stride, slope, out_f32)


# ---------------------------------------------------------------------------------------------------------------
# BatchNorm2d (+ residual + LeakyReLU) on token maps
# ---------------------------------------------------------------------------------------------------------------
class BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, nbt, res, training, slope):
        rows, C = x.shape
        dev = x.device
        sums, mr = _zeros((2, C), dev), torch.empty((2, C), dtype=torch.float32, device=dev)
        y = torch.empty((rows, C), dtype=x.dtype, device=dev)
        call('fw_bn_cl_fwd', dt(x.dtype), x, x.stride(0), gamma, beta, rmean, rvar, nbt, sums, mr, res, res.stride(0) if res is not None else 0,
             y, C, rows, C, int(training), 1e-5, 0.1, float(slope))
        ctx.save_for_backward(x, gamma, mr, y)
        ctx.cfg = (training, slope, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mr, y = ctx.saved_tensors
        training, slope, has_res = ctx.cfg
        rows, C = x.shape
        dy = dy.contiguous()
        sums = _zeros((2, C), x.device)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        call('fw_bn_cl_bwd', dt(x.dtype), dy, dy.stride(0), y, y.stride(0), x, x.stride(0), mr, gamma, sums, dx, dx.stride(0), dres,
             dres.stride(0) if dres is not None else 0, rows, C, int(training), float(slope))
        return dx, sums[1], sums[0], None, None, None, dres, None, None


def batch_norm(x, bn, training, slope=1.0, res=None):
    return BatchNormFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, res, training, slope)


class LreluTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        y = torch.empty((x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
        call('fw_lrelu_t', dt(x.dtype), x, x.stride(0), y, y.stride(0), x.shape[0], x.shape[1], float(slope))
        ctx.save_for_backward(y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        call('fw_lrelu_t_bwd', dt(y.dtype), dy, dy.stride(0), y, y.stride(0), dx, dx.stride(0), y.shape[0], y.shape[1], float(ctx.slope))
        return dx, None


class GapFn(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) + squeeze: T [B*P, C] -> f32 [B, C]."""

    @staticmethod
    def forward(ctx, x, B):
        rows, C = x.shape
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        call('fw_gap_cl', dt(x.dtype), x, x.stride(0), out, B, rows // B, C)
        ctx.geo = (B, rows // B, C, x.dtype)
        return out

    @staticmethod
    def backward(ctx, dgap):
        B, P, C, dtype = ctx.geo
        dx = torch.empty((B * P, C), dtype=dtype, device=dgap.device)
        call('fw_gap_cl_bwd', dt(dtype), dgap.contiguous().float(), dx, C, B, P, C)
        return dx, None


# ---------------------------------------------------------------------------------------------------------------
# ResNet encoder  (net/encoder_ResNet.py)
# ---------------------------------------------------------------------------------------------------------------
class ResBlock(nn.Module):
    """encoder_ResNet.py:4-20."""

    def __init__(self, in_feat, out_feat, stride=1):
        super().__init__()
        self.stride = stride
        self.backbone = nn.Sequential(
            nn.Conv2d(in_feat, out_feat, kernel_size=3, stride=stride, padding=1, bias=False), nn.BatchNorm2d(out_feat), nn.LeakyReLU(0.1, True),
            nn.Conv2d(out_feat, out_feat, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(out_feat))
        self.shortcut = nn.Sequential(nn.Conv2d(in_feat, out_feat, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(out_feat))

    def run(self, x, geo):
        """x: T tokens of a [B, H, W] map -> (tokens of the output map, its geometry)."""
        B, H, W = geo
        s = self.stride
        go = (B, (H - 1) // s + 1, (W - 1) // s + 1)
        y = conv(x, self.backbone[0], geo, s)
        y = batch_norm(y, self.backbone[1], self.training, 0.1)
        y = conv(y, self.backbone[3], go, 1)
        sc = conv(x, self.shortcut[0], geo, s)
        sc = batch_norm(sc, self.shortcut[1], self.training)
        return batch_norm(y, self.backbone[4], self.training, 0.1, res=sc), go          # LReLU(0.1)(backbone + shortcut), :20

    def forward(self, x):                                       # API parity: NCHW f32 in / out
        B, C, H, W = x.shape
        y, (_, Ho, Wo) = self.run(TokensFn.apply(x), (B, H, W))
        return ToImageFn.apply(y, B, y.shape[1], Ho, Wo)


def nchw_view(tok, B, H, W):
    """The reference's [B, C, H, W] shape as a VIEW of the token-major buffer (no copy; `tokens_of` undoes it)."""
    return tok.view(B, H, W, tok.shape[1]).permute(0, 3, 1, 2)


def tokens_of(t):
    """[B, C, H, W] tensor -> (token-major T map [B*H*W, C], (B, H, W)); free for the channels-last views `nchw_view` hands out."""
    B, C, H, W = t.shape
    tok = t.permute(0, 2, 3, 1)
    want = Fn.config.compute_dtype
    if tok.dtype != want or not tok.is_contiguous() or (C * tok.element_size()) % 16:
        if (C * (2 if want == torch.bfloat16 else 4)) % 16 == 0:
            tok = tok.contiguous().to(want)
        else:
            return TokensFn.apply(t), (B, H, W)
    return tok.reshape(B * H * W, C), (B, H, W)


class ResNetEncoder(nn.Module):
    """encoder_ResNet.py:23-47.  forward(x) -> (fea [B, dim], [out], inter [B, dim/4, H, W])."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.dim = opt.encoder_dim
        self.E_pre = ResBlock(in_feat=3, out_feat=self.dim // 4, stride=1)
        self.E = nn.Sequential(ResBlock(in_feat=self.dim // 4, out_feat=self.dim // 2, stride=2),
                               ResBlock(in_feat=self.dim // 2, out_feat=self.dim, stride=2), nn.AdaptiveAvgPool2d(1))
        self.mlp = nn.Sequential(nn.Linear(self.dim, self.dim), nn.LeakyReLU(0.1, True), nn.Linear(self.dim, self.dim))

    def forward(self, x, want_heads=True):
        B, _, H, W = x.shape
        inter, geo = self.E_pre.run(TokensFn.apply(x.contiguous().float()), (B, H, W))
        y, g1 = self.E[0].run(inter, geo)
        y, g2 = self.E[1].run(y, g1)
        fea = GapFn.apply(y, B)                                                  # :33,44
        out = []
        if want_heads:
            h = Fn.linear(Fn.CastFn.apply(fea), self.mlp[0].weight, self.mlp[0].bias, out_f32=True)
            out = [Fn.linear(Fn.LreluFn.apply(h, 0.1), self.mlp[2].weight, self.mlp[2].bias, out_f32=True)]
        return fea, out, nchw_view(inter, B, H, W)


# ---------------------------------------------------------------------------------------------------------------
# DGRN decoder  (net/decoder_DGRN.py, net/utils/deform_conv.py)
# ---------------------------------------------------------------------------------------------------------------
class DcnFn(torch.autograd.Function):
    """Modulated deformable 3x3 convolution (DCNv2) of x with offsets / mask logits `om` (f32 [M, 32]); parity unpinned."""

    @staticmethod
    def forward(ctx, x, om, weight, geo):
        B, H, W = geo
        Co, Ci = weight.shape[0], weight.shape[1]
        M = x.shape[0]
        col = torch.empty((M, 9 * Ci), dtype=x.dtype, device=x.device)
        call('fw_dcn_im2col', dt(x.dtype), x, x.stride(0), om, col, B, H, W, Ci)
        w = panel(weight, 'fwd')                                                 # [Co][9 * Ci] tap-major = the GEMM's W
        y = ops.gemm(col, w[:Co], M, Co, 9 * Ci)
        ctx.save_for_backward(x, om, weight)
        ctx.geo = geo
        return y

    @staticmethod
    def backward(ctx, dy):
        x, om, weight = ctx.saved_tensors
        B, H, W = ctx.geo
        Co, Ci = weight.shape[0], weight.shape[1]
        M = x.shape[0]
        dy = Fn.aligned(dy.contiguous())
        col = torch.empty((M, 9 * Ci), dtype=x.dtype, device=x.device)           # recomputed: 9x the activation, not kept across the pass
        call('fw_dcn_im2col', dt(x.dtype), x, x.stride(0), om, col, B, H, W, Ci)
        dwk = _zeros((Co, 9 * Ci), x.device)
        ops.wgrad(dy, col, Co, 9 * Ci, M, dwk)
        dw = dwk.view(Co, 9, Ci).permute(0, 2, 1).reshape(Co, Ci, 3, 3).contiguous()
        w = panel(weight, 'fwd')
        dcol = ops.gemm(dy, w[:Co], M, 9 * Ci, Co, w_trans=True)
        dxf = _zeros((M, Ci), x.device)
        dom = _zeros((M, 32), x.device)
        call('fw_dcn_bwd', dt(x.dtype), dcol, x, x.stride(0), om, dxf, Ci, dom, B, H, W, Ci)
        dx = dxf if x.dtype == torch.float32 else ops.cast_rows(dxf, x.dtype)
        return dx, dom, dw, None


class DCN_layer(nn.Module):
    """net/utils/deform_conv.py:10-67 (the constructor and initialisation are the reference's; forward = DCNv2, see the module header)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, bias=True,
                 extra_offset_mask=True):
        super().__init__()
        assert kernel_size == 3 and stride == 1 and padding == 1 and dilation == 1 and groups == 1 and deformable_groups == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels, 3, 3))
        self.conv_offset_mask = nn.Conv2d(in_channels * 2, 27, kernel_size=3, stride=1, padding=1, bias=True)
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.conv_offset_mask.weight.data.zero_()                                 # :52-54
        self.conv_offset_mask.bias.data.zero_()
        stdv = 1. / (in_channels * 9) ** 0.5                                      # :43-50
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()

    def run(self, x, inter, geo):
        # :57-62 -- conv_offset_mask on cat[x, inter]; (o1 | o2) re-joined in order are the 18 offset channels, the last 9 the mask logits
        om = conv(x, self.conv_offset_mask, geo, x2=inter, out_f32=True)          # f32 [M, 32]
        y = DcnFn.apply(x, om, self.weight, geo)
        assert self.bias is None, 'DGM builds its DCN without bias (decoder_DGRN.py:16-17)'
        return y


class SFT_layer(nn.Module):
    """decoder_DGRN.py:35-57: x * gamma(inter) + beta(inter); the product and sum are fused into DGM's combine kernel."""

    def __init__(self, channels_in, channels_out):
        super().__init__()
        mk = lambda: nn.Sequential(nn.Conv2d(channels_in, channels_out, 1, 1, 0, bias=False), nn.LeakyReLU(0.1, True),
                                   nn.Conv2d(channels_out, channels_out, 1, 1, 0, bias=False))
        self.conv_gamma, self.conv_beta = mk(), mk()

    def gamma_beta(self, inter):
        def mlp(seq):
            h = LreluTFn.apply(Fn.linear(inter, seq[0].weight), 0.1)
            return Fn.linear(h, seq[2].weight)
        return mlp(self.conv_gamma), mlp(self.conv_beta)


class DgmFn(torch.autograd.Function):
    """lrelu(x + dcn + x * gamma + beta, slope)  (decoder_DGRN.py:28-32 and the LeakyReLU DGB applies next, :79,81)."""

    @staticmethod
    def forward(ctx, x, dcn, gamma, beta, slope):
        x, dcn, gamma, beta = (t.contiguous() for t in (x, dcn, gamma, beta))
        out = torch.empty_like(x)
        call('fw_dgm_fwd', dt(x.dtype), x, dcn, gamma, beta, out, x.numel(), float(slope))
        ctx.save_for_backward(x, gamma, out)
        ctx.slope = slope
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, out = ctx.saved_tensors
        dout = dout.contiguous()
        dx, dz, dg = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        call('fw_dgm_bwd', dt(x.dtype), dout, out, x, gamma, dx, dz, dg, x.numel(), float(ctx.slope))
        return dx, dz, dg, dz, None


class DGM(nn.Module):
    """decoder_DGRN.py:9-32."""

    def __init__(self, channels_in, channels_out, kernel_size):
        super().__init__()
        self.dcn = DCN_layer(channels_in, channels_out, kernel_size, padding=(kernel_size - 1) // 2, bias=False)
        self.sft = SFT_layer(channels_in, channels_out)
        self.relu = nn.LeakyReLU(0.1, True)

    def run(self, x, inter, geo, slope=1.0):
        gamma, beta = self.sft.gamma_beta(inter)
        return DgmFn.apply(x, self.dcn.run(x, inter, geo), gamma, beta, slope)


class DGB(nn.Module):
    """decoder_DGRN.py:60-84."""

    def __init__(self, conv_, n_feat, kernel_size):
        super().__init__()
        self.dgm1, self.dgm2 = DGM(n_feat, n_feat, kernel_size), DGM(n_feat, n_feat, kernel_size)
        self.conv1, self.conv2 = conv_(n_feat, n_feat, kernel_size), conv_(n_feat, n_feat, kernel_size)
        self.relu = nn.LeakyReLU(0.1, True)

    def run(self, x, inter, geo):
        out = self.dgm1.run(x, inter, geo, 0.1)
        out = conv(out, self.conv1, geo, slope=0.1)
        out = self.dgm2.run(out, inter, geo, 0.1)
        return conv(out, self.conv2, geo, res=x)


def default_conv(in_channels, out_channels, kernel_size, bias=True):
    return nn.Conv2d(in_channels, out_channels, kernel_size, padding=(kernel_size // 2), bias=bias)


class DGG(nn.Module):
    """decoder_DGRN.py:87-110."""

    def __init__(self, conv_, n_feat, kernel_size, n_blocks):
        super().__init__()
        self.n_blocks = n_blocks
        self.body = nn.Sequential(*([DGB(conv_, n_feat, kernel_size) for _ in range(n_blocks)] + [conv_(n_feat, n_feat, kernel_size)]))

    def run(self, x, inter, geo):
        res = x
        for i in range(self.n_blocks):
            res = self.body[i].run(res, inter, geo)
        return conv(res, self.body[-1], geo, res=x)


class DGRN(nn.Module):
    """decoder_DGRN.py:113-158.  forward(x [B, 3, H, W], inter [B, n_feats, H, W]) -> [B, 3, H, W] (no global residual)."""

    def __init__(self, opt, conv_=default_conv):
        super().__init__()
        self.n_groups = 5
        n_blocks = 5
        if opt.encoder_type == 'ResNet':
            n_feats = opt.encoder_dim // 4
        elif opt.encoder_type == 'ViT':
            n_feats = opt.encoder_dim
        else:
            raise NotImplementedError('DGRN pairs with the ResNet or ViT encoder (decoder_DGRN.py:120-124)')
        if n_feats % 64:
            raise NotImplementedError(f'the implicit-GEMM convolutions stage 64 channels at a time; n_feats = {n_feats}')
        self.n_feats = n_feats
        self.head = nn.Sequential(conv_(3, n_feats, 3))
        self.body = nn.Sequential(*([DGG(default_conv, n_feats, 3, n_blocks) for _ in range(self.n_groups)] + [conv_(n_feats, n_feats, 3)]))
        self.tail = nn.Sequential(conv_(n_feats, 3, 3))

    def forward(self, x, inter):
        B, _, H, W = x.shape
        geo = (B, H, W)
        it, _ = tokens_of(inter)
        h = conv(TokensFn.apply(x.contiguous().float()), self.head[0], geo)
        res = h
        for i in range(self.n_groups):
            res = self.body[i].run(res, it, geo)
        res = conv(res, self.body[-1], geo, res=h)
        y = conv(res, self.tail[0], geo)                                         # T [M, 8], channels 0..2
        return ToImageFn.apply(y, B, 3, H, W)
