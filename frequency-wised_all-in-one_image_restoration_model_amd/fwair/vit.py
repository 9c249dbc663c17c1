"""`ViTEncoder` -- the reference's ViT plug-in of the encoder seam (net/encoder_ViT.py:17-203, BASELINE configs[4]) on the HIP
kernels: same class / attribute names (`to_patch_embedding`, `pos_embedding`, `transformer.layers[i][0].fn`, `mlp_head`, `norm`, `avg`,
`mlp`, `depth`), constructor `(opt)`, return value `(fea, [out], inter)` and state_dict keys (tests/golden/schema.json
`vit_uformer`, dumped from the reference).

Token stream: f32 [B*N, 768] (the residual stream, as in the Uformer blocks); GEMM / attention operands T.  One image is one
attention "window" of N = (S/16)^2 tokens: at 128x128 N = 64 and the window-attention kernel runs it as a single 8x8 window per
image with head_dim 64, 12 heads and a zero relative-position table.

Not built (raise NotImplementedError instead of diverging silently):
  * image_size != 128: N = 256 tokens need a multi-tile global-attention kernel; the reference cannot construct it either
    (pos_embedding and the band masks are sized for 128);
  * `frequency_decompose_type` != 'none' (the learnable band re-weighting `lamb` of encoder_ViT.py:56-68,86-92; the option's default
    is 'none').
Dropout (p = 0.1 in train mode, encoder_ViT.py:128-129) is not applied: the goldens neutralise it, and the only end-to-end
configuration of the reference that uses this encoder runs in eval mode.
"""
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


class AddPosFn(torch.autograd.Function):
    """x [B*N, C] + pos_embedding[:, :N]  (encoder_ViT.py:187); d(pos) = sum over the batch."""

    @staticmethod
    def forward(ctx, x, pos, B):
        x = x.contiguous()
        n = x.shape[0] // B
        p = pos[0, :n].contiguous()
        out = torch.empty_like(x)
        call('fw_add_bcast', x, p, out, x.numel(), p.numel())
        ctx.geo = (B, n, pos.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, n, pshape = ctx.geo
        dy = dy.contiguous()
        dpos = torch.zeros(pshape, dtype=torch.float32, device=dy.device)
        acc = torch.zeros(n * dy.shape[1], dtype=torch.float32, device=dy.device)
        ops.colsum(dy.view(B, n * dy.shape[1]), acc)
        dpos[0, :n] = acc.view(n, dy.shape[1])
        return dy, dpos, None


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


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0., decompose_type='none', wised_batch=None):
        super().__init__()
        if decompose_type != 'none':
            raise NotImplementedError("ViT band re-weighting (frequency_decompose_type != 'none', encoder_ViT.py:56-68) is not built")
        if dim_head != 64:
            raise NotImplementedError(f'the global-attention path is instantiated for head_dim 64, not {dim_head}')
        inner = dim_head * heads
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.num_bands = None
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))
        self.register_buffer('_zero_table', torch.zeros(1, 225, heads), persistent=False)


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0., decompose_type='none', wised_batch=None):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([
            PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout, decompose_type=decompose_type, wised_batch=wised_batch)),
            PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout))]) for _ in range(depth)])

    def run(self, x, B):
        """x: f32 [B*N, dim] -> same (encoder_ViT.py:112-116: x = attn(x) + x; x = ff(x) + x, both pre-norm)."""
        N = x.shape[0] // B
        side = int(round(N ** 0.5))
        for attn, ff in self.layers:
            a = attn.fn
            x, xn = Fn.LnResFn.apply(x, attn.norm.weight, attn.norm.bias)
            qkv = Fn.linear(xn, a.to_qkv.weight)                                              # [B*N, 3 * inner] = q | k | v, head h at column h*64
            C = a.heads * a.dim_head
            geo = (C, B, side, side, a.heads, 1, 0, 0, 0)                                     # one 8x8 "window" per image
            o = Fn.WindowAttnFn.apply(qkv, a._zero_table, None, geo, None, None)
            x = Fn.linear(o, a.to_out[0].weight, a.to_out[0].bias, residual=x)
            x, xn = Fn.LnResFn.apply(x, ff.norm.weight, ff.norm.bias)
            h, g = Fn.linear(xn, ff.fn.net[0].weight, ff.fn.net[0].bias, gelu_out=True)
            x = Fn.linear(g, ff.fn.net[3].weight, ff.fn.net[3].bias, residual=x, x_pre=h)
        return x


class ViTEncoder(nn.Module):
    """encoder_ViT.py:119-203."""

    def __init__(self, opt, image_size=128, patch_size=16, depth=12, heads=12, mlp_dim=3072, channels=3, dropout=0.1, emb_dropout=0.1):
        super().__init__()
        out_channels = opt.out_channels
        dim = out_channels * patch_size * patch_size
        self.opt, self.depth = opt, depth
        dim_head = dim // heads
        if getattr(opt, 'patch_size', 128) not in (None, 128) or image_size != 128:
            raise NotImplementedError('ViTEncoder is built for 128x128 inputs (64 tokens = one attention tile); the reference cannot '
                                      'construct another size either (pos_embedding / band masks, SURVEY.md 8a row a20)')
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
        t = AddPosFn.apply(t, self.pos_embedding, B)
        t = self.transformer.run(t, B)
        t = Fn.LayerNormFn.apply(t, self.mlp_head[0].weight, self.mlp_head[0].bias)
        fmap = Fn.linear(t, self.mlp_head[1].weight, self.mlp_head[1].bias)                 # T [B*N, 256 * encoder_dim] = planes [B][ED][H*W]
        bn = self.norm[0]
        inter, fea = BnPlanesFn.apply(fmap.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, B, self.training)
        out = []
        if want_heads:
            h = SmallLinearFn.apply(fea, self.mlp[0].weight, self.mlp[0].bias, 0.1)
            out = [SmallLinearFn.apply(h, self.mlp[2].weight, self.mlp[2].bias, 1.0)]
        return fea, out, inter.view(B, self.opt.encoder_dim, H, W)
