"""`ViTEncoder` -- the reference's ViT plug-in of the encoder seam (net/encoder_ViT.py:17-203, BASELINE configs[4]) on the HIP
kernels: same class / attribute names (`to_patch_embedding`, `pos_embedding`, `transformer.layers[i][0].fn`, `mlp_head`, `norm`, `avg`,
`mlp`, `depth`), constructor `(opt)`, return value `(fea, [out], inter)` and state_dict keys (tests/golden/schema.json
`vit_uformer`, dumped from the reference).

Token stream: f32 [B*N, 768] (the residual stream, as in the Uformer blocks); GEMM / attention operands T.  N = (S/16)^2 tokens per
image attend globally (csrc/fw_gattn.hip): N = 64 at 128x128, N = 256 at 256x256 (BASELINE configs[4]).  The image side follows
`opt.patch_size` like the Uformer classes (the reference's seam passes only `opt` and so always builds for 128, net/model.py:31).

Train mode applies every nn.Dropout of the reference (p = 0.1: embedding, attention map, to_out, FeedForward hidden and output;
encoder_ViT.py:31,33,67,73,158) as counter-based masks: mask = f(seed, call site, element index), re-derived in the backward
pass, reproduced bit for bit by the CPU oracle (oracle/dropout_hash.py).  The STREAM of random numbers differs from torch's Philox
stream -- as it does between torch's own CPU and GPU generators -- the Bernoulli(1 - p) / (1 - p) semantics are the reference's.

`frequency_decompose_type` in {'DC', '<n>_bands'}: the learnable band re-weighting `lamb` of encoder_ViT.py:51-66,85-92 runs inside
the attention kernel (64x64 2-D DFT on the f32 MFMA).  Like the reference it needs N = dim_head = 64, i.e. 128x128 inputs (its masks
are dim_head x dim_head, :56,60); at any other size both raise.
"""
import math
import zlib

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .lib import call, dt


class LayerNormF32Fn(torch.autograd.Function):
    """nn.LayerNorm with an f32 result (the patch embedding's second norm feeds the f32 token stream)."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, torch.float32)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.beta = beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dg, rg = Fn._grad_target(gamma)
        db, rb = Fn._grad_target(ctx.beta)
        dx = ops.layernorm_bwd(dy.contiguous().float(), x, gamma, mean, rstd, dg, db, defer=rg is None and rb is None)
        return dx, rg, rb


class BnPlanesFn(torch.autograd.Function):
    """fea T [B*N, ED*P/N] viewed as planes [B][ED][P] -> (inter f32 [B, ED, P] = LeakyReLU(BatchNorm2d), gap f32 [B, ED])."""

    @staticmethod
    def forward(ctx, fea, gamma, beta, rmean, rvar, nbt, B, training):
        ED = gamma.shape[0]
        P = fea.numel() // (B * ED)
        assert fea.is_contiguous()
        dev = fea.device
        mr = torch.empty((2, ED), dtype=torch.float32, device=dev)
        inter = torch.empty((B, ED, P), dtype=torch.float32, device=dev)
        gap = torch.empty((B, ED), dtype=torch.float32, device=dev)
        call('fw_bn_planes_fwd', dt(fea.dtype), fea, gamma, beta, rmean, rvar, nbt, mr, inter, gap, B, ED, P, int(training), 1e-5, 0.1, 0.1)
        ctx.save_for_backward(fea, gamma, mr, inter)
        ctx.geo = (B, ED, P, training)
        ctx.set_materialize_grads(False)
        return inter, gap

    @staticmethod
    def backward(ctx, dinter, dgap):
        fea, gamma, mr, inter = ctx.saved_tensors
        B, ED, P, training = ctx.geo
        if dinter is None and dgap is None:
            return (None,) * 8
        dev = fea.device
        dfea = torch.empty_like(fea)
        dg, db = torch.empty(ED, dtype=torch.float32, device=dev), torch.empty(ED, dtype=torch.float32, device=dev)
        call('fw_bn_planes_bwd', dt(fea.dtype), fea, inter, gamma, mr, dinter.contiguous().float() if dinter is not None else None,
             dgap.contiguous().float() if dgap is not None else None, dfea, dg, db, B, ED, P, int(training), 0.1)
        return dfea, dg, db, None, None, None, None, None


class SmallLinearFn(torch.autograd.Function):
    """y = lrelu(x W^T + b, slope) for the tiny head MLP (f32 throughout)."""

    @staticmethod
    def forward(ctx, x, weight, bias, slope):
        x = x.contiguous().float()
        M, K = x.shape
        N = weight.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        call('fw_small_linear_fwd', x, weight, bias, y, M, N, K, float(slope))
        ctx.save_for_backward(x, weight, y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        M, K = x.shape
        N = weight.shape[0]
        dx, dw, db = torch.empty_like(x), torch.empty_like(weight), torch.empty(N, dtype=torch.float32, device=x.device)
        call('fw_small_linear_bwd', dy.contiguous().float(), y, x, weight, dx, dw, db, M, N, K, float(ctx.slope))
        return dx, dw, db, None


# ---------------------------------------------------------------------------------------------------------------
# module tree (names = the reference's, for the state_dict)
# ---------------------------------------------------------------------------------------------------------------
class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden_dim, dim), nn.Dropout(dropout))


_consts = {}


def _spectral_tables(kind, nb, device):
    """(bandidx u8 [64][64] in un-shifted coordinates, f32 cos | sin panels [2][64][64]) of the 64x64 attention-map decomposition
    (encoder_ViT.py:53-60: FrequencyDecompose('frequency_decompose', 1/nb, 64, 64) or 'frequency_decompose_dc')."""
    from . import lfs
    key = (kind, nb, str(device))
    if key not in _consts:
        if kind == 'DC':
            idx = torch.ones((64, 64), dtype=torch.uint8)
            idx[0, 0] = 0                                                # band 0 = the mean (DC bin), band 1 = everything else
        else:
            masks = lfs.band_masks_shifted('frequency_decompose', 1. / nb, 64, 64)
            shifted = torch.zeros((64, 64), dtype=torch.uint8)
            for i, m in enumerate(masks):
                shifted[m] = i
            assert bool(torch.stack(masks).sum(0).eq(1).all())
            idx = torch.fft.ifftshift(shifted, dim=(0, 1)).contiguous()
        assert bool((idx == idx.t()).all())                              # radial masks: the filter commutes with the transpose
        ang = 2 * math.pi * torch.outer(torch.arange(64, dtype=torch.float64), torch.arange(64, dtype=torch.float64)) / 64
        panels = torch.stack([torch.cos(ang), torch.sin(ang)]).float().contiguous()
        _consts[key] = (idx.to(device), panels.to(device))
    return _consts[key]


class GlobalAttnFn(torch.autograd.Function):
    """qkv: T [B*N, 3*heads*64] (q | k | v) -> out T [B*N, heads*64]  (encoder_ViT.py:76-96 between to_qkv and to_out)."""

    @staticmethod
    def forward(ctx, qkv, lamb, meta):
        B, N, heads, p, site, spec = meta
        inner = heads * 64
        out = Fn.act_empty(qkv.shape[0], inner, qkv.dtype, qkv.device)
        lse = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
        seed = Fn.dropout_seed(qkv.device) if p > 0 else None
        bidx, panels = spec if spec is not None else (None, None)
        nb, lb = (lamb.shape[0], lamb.shape[1]) if lamb is not None else (0, 1)
        lam = lamb.detach().contiguous() if lamb is not None else None
        call('fw_gattn_fwd', dt(qkv.dtype), qkv, qkv[:, inner:], qkv[:, 2 * inner:], qkv.stride(0), out, out.stride(0), lse, B, heads, N,
             64 ** -0.5, seed, site, float(p), lam, nb, lb, bidx, panels)
        ctx.save_for_backward(qkv, out, lse, lam)
        ctx.meta = (meta, seed)
        ctx.lamb_param = lamb
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, lam = ctx.saved_tensors
        (B, N, heads, p, site, spec), seed = ctx.meta
        inner = heads * 64
        dout = Fn.aligned(dout)
        dqkv = Fn.act_empty(qkv.shape[0], qkv.shape[1], qkv.dtype, qkv.device)
        bidx, panels = spec if spec is not None else (None, None)
        dlam = rl = None
        dvec = None
        if lam is not None:
            dlam, rl = Fn._grad_target(ctx.lamb_param)
        else:
            dvec = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
        nb, lb = (lam.shape[0], lam.shape[1]) if lam is not None else (0, 1)
        call('fw_gattn_bwd', dt(qkv.dtype), qkv, qkv[:, inner:], qkv[:, 2 * inner:], qkv.stride(0), out, out.stride(0), dout, dout.stride(0),
             lse, dvec, dqkv, dqkv[:, inner:], dqkv[:, 2 * inner:], dqkv.stride(0), B, heads, N, 64 ** -0.5, seed, site, float(p),
             lam, dlam, nb, lb, bidx, panels)
        return dqkv, rl, None


class EmbDropFn(torch.autograd.Function):
    """dropout(x + pos_embedding[:, :N])  (encoder_ViT.py:187-189); x f32 [B*N, C]."""

    @staticmethod
    def forward(ctx, x, pos, B, site, p):
        x = x.contiguous()
        n = x.shape[0] // B
        pe = pos[0, :n].contiguous()
        out = torch.empty_like(x)
        seed = Fn.dropout_seed(x.device) if p > 0 else None
        call('fw_dropout', 4, 0, x, pe, None, out, x.numel(), pe.numel(), seed, site, float(p))
        ctx.geo = (B, n, pos.shape, site, p, seed)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, n, pshape, site, p, seed = ctx.geo
        dy = dy.contiguous()
        if p > 0:
            dx = torch.empty_like(dy)
            call('fw_dropout', 0, 0, dy, None, None, dx, dy.numel(), 1, seed, site, float(p))
        else:
            dx = dy
        dpos = torch.zeros(pshape, dtype=torch.float32, device=dy.device)
        acc = torch.zeros(n * dy.shape[1], dtype=torch.float32, device=dy.device)
        ops.colsum(dx.view(B, n * dy.shape[1]), acc)
        dpos[0, :n] = acc.view(n, dy.shape[1])
        return dx, dpos, None, None, None


class Attention(nn.Module):
    """encoder_ViT.py:38-98."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0., decompose_type='none', wised_batch=None):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError(f'the global-attention kernel is instantiated for head_dim 64, not {dim_head}')
        inner = dim_head * heads
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.num_bands = None
        self._kind = None
        if decompose_type != 'none':                                     # :51-66
            if decompose_type.split('_')[-1] == 'bands':
                self.num_bands = int(decompose_type.split('_')[0])
                self._kind = 'bands'
            elif decompose_type == 'DC':
                self.num_bands = 2
                self._kind = 'DC'
            if self.num_bands is not None and not 1 <= self.num_bands <= 16:
                raise NotImplementedError('band re-weighting: 1..16 bands')
            self.lamb = nn.Parameter(torch.zeros(self.num_bands, 1 if wised_batch is None else wised_batch, heads))
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0., decompose_type='none', wised_batch=None):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([
            PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout, decompose_type=decompose_type, wised_batch=wised_batch)),
            PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout))]) for _ in range(depth)])

    def run(self, x, B, site_base=0, training=False):
        """x: f32 [B*N, dim] -> same (encoder_ViT.py:112-116: x = attn(x) + x; x = ff(x) + x, both pre-norm).
        Dropout call sites of layer i: site_base + 4 i + {0: attention map, 1: to_out, 2: FeedForward hidden, 3: FeedForward output}."""
        N = x.shape[0] // B
        for li, (attn, ff) in enumerate(self.layers):
            a = attn.fn
            pa = float(a.dropout.p) if training else 0.0
            po = float(a.to_out[1].p) if training else 0.0
            ph = float(ff.fn.net[2].p) if training else 0.0
            pf = float(ff.fn.net[4].p) if training else 0.0
            site = site_base + 4 * li
            x, xn = Fn.LnResFn.apply(x, attn.norm.weight, attn.norm.bias)
            qkv = Fn.linear(xn, a.to_qkv.weight)                                              # [B*N, 3 * inner] = q | k | v, head h at column h*64
            lamb, spec = None, None
            if a.num_bands is not None:
                if N != 64:
                    raise NotImplementedError('the band re-weighting needs N = dim_head = 64 tokens (128x128 inputs): the reference sizes '
                                              'its masks dim_head x dim_head (encoder_ViT.py:56,60) and fails otherwise too')
                if a.lamb.shape[1] not in (1, B):
                    raise NotImplementedError(f'batch-wise lamb was built for batch {a.lamb.shape[1]}, got {B}')
                lamb, spec = a.lamb, _spectral_tables(a._kind, a.num_bands, x.device)
            o = GlobalAttnFn.apply(qkv, lamb, (B, N, a.heads, pa, site, spec))
            if po > 0:
                y = Fn.linear(o, a.to_out[0].weight, a.to_out[0].bias, out_f32=True)
                x = Fn.DropAddFn.apply(y, x, site + 1, po)
            else:
                x = Fn.linear(o, a.to_out[0].weight, a.to_out[0].bias, residual=x)
            x, xn = Fn.LnResFn.apply(x, ff.norm.weight, ff.norm.bias)
            if ph > 0 or pf > 0:
                h = Fn.linear(xn, ff.fn.net[0].weight, ff.fn.net[0].bias)
                g = Fn.GeluDropFn.apply(h.contiguous(), site + 2, ph)
                y = Fn.linear(g, ff.fn.net[3].weight, ff.fn.net[3].bias, out_f32=True)
                x = Fn.DropAddFn.apply(y, x, site + 3, pf)
            else:
                h, g = Fn.linear(xn, ff.fn.net[0].weight, ff.fn.net[0].bias, gelu_out=True)
                x = Fn.linear(g, ff.fn.net[3].weight, ff.fn.net[3].bias, residual=x, x_pre=h)
        return x


class ViTEncoder(nn.Module):
    """encoder_ViT.py:119-203."""

    def __init__(self, opt, image_size=None, patch_size=16, depth=12, heads=12, mlp_dim=3072, channels=3, dropout=0.1, emb_dropout=0.1):
        super().__init__()
        out_channels = opt.out_channels
        dim = out_channels * patch_size * patch_size
        self.opt, self.depth = opt, depth
        dim_head = dim // heads
        if image_size is None:                                           # the seam passes only `opt` (net/model.py:31): follow --patch_size
            image_size = int(getattr(opt, 'patch_size', None) or 128)
        if image_size not in (128, 256):
            raise NotImplementedError(f'ViTEncoder: image_size {image_size}: the global-attention kernel holds N = (S/16)^2 in {{64, 256}} keys per head')
        self.image_height = self.image_width = image_size
        self.patch = patch_size
        num_patches = (image_size // patch_size) ** 2
        patch_dim = channels * patch_size * patch_size
        self.to_patch_embedding = nn.Sequential(nn.Identity(), nn.LayerNorm(patch_dim), nn.Linear(patch_dim, dim), nn.LayerNorm(dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout, decompose_type=opt.frequency_decompose_type,
                                       wised_batch=opt.batch_size if getattr(opt, 'batch_wise_decompose', False) else None)
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim // out_channels * opt.encoder_dim))
        self.norm = nn.Sequential(nn.BatchNorm2d(opt.encoder_dim), nn.LeakyReLU(0.1, True))
        self.avg = nn.AdaptiveAvgPool2d(1)
        self.mlp = nn.Sequential(nn.Linear(opt.encoder_dim, opt.encoder_dim), nn.LeakyReLU(0.1, True), nn.Linear(opt.encoder_dim, opt.encoder_dim))
        self.set_prefix('')

    def set_prefix(self, prefix):
        """MoCo names its two copies (E.E.encoder_q. / E.E.encoder_k.): their Dropout call sites must draw different masks."""
        self._site_base = ((zlib.crc32(prefix.encode()) & 0xFFFF) << 8) if prefix else 0

    def forward(self, x, want_heads=True):
        B, C, H, W = x.shape
        if H != self.image_height or W != self.image_width:
            raise NotImplementedError(f'the ViT encoder is built for {self.image_height}x{self.image_width} inputs (encoder_ViT.py:122)')
        p = self.patch
        hh, ww = H // p, W // p
        # 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (encoder_ViT.py:150): pure data movement
        t = x.float().reshape(B, C, hh, p, ww, p).permute(0, 2, 4, 3, 5, 1).reshape(B * hh * ww, p * p * C).contiguous()
        e = self.to_patch_embedding
        t = Fn.LayerNormFn.apply(t, e[1].weight, e[1].bias)
        t = Fn.linear(t, e[2].weight, e[2].bias, out_f32=True)
        t = LayerNormF32Fn.apply(t, e[3].weight, e[3].bias)
        t = EmbDropFn.apply(t, self.pos_embedding, B, self._site_base + 0xFF, float(self.dropout.p) if self.training else 0.0)
        t = self.transformer.run(t, B, self._site_base, self.training)
        t = Fn.LayerNormFn.apply(t, self.mlp_head[0].weight, self.mlp_head[0].bias)
        fmap = Fn.linear(t, self.mlp_head[1].weight, self.mlp_head[1].bias)                 # T [B*N, 256 * encoder_dim] = planes [B][ED][H*W]
        bn = self.norm[0]
        inter, fea = BnPlanesFn.apply(fmap.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, B, self.training)
        out = []
        if want_heads:
            h = SmallLinearFn.apply(fea, self.mlp[0].weight, self.mlp[0].bias, 0.1)
            out = [SmallLinearFn.apply(h, self.mlp[2].weight, self.mlp[2].bias, 1.0)]
        return fea, out, inter.view(B, self.opt.encoder_dim, H, W)
