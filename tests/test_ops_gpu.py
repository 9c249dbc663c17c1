"""GPU parity of every C-ABI kernel against the CPU oracle math (torch fp64/fp32 on the host).

Tolerances (max abs err / max abs ref):  f32 kernels 5e-5 (exact-f32 MFMA chains, order of summation
differs), bf16 kernels 2e-2 (operands rounded to 8 significant bits, f32 accumulation).
Every call goes through fwair.lib.call -> libfwair_hip.so; nothing here has a CPU implementation."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import airnet_oracle as O

pytestmark = pytest.mark.gpu

DEV = 'cuda'
TOL = {torch.float32: 5e-5, torch.bfloat16: 2e-2}
DTYPES = [torch.float32, torch.bfloat16]


def ops():
    from fwair import ops as _ops
    return _ops


def call(*a):
    from fwair.lib import call as _c
    return _c(*a)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g, dtype=torch.float32) * scale


def close(a, b, tol, what=''):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, f'{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}'
    assert torch.isfinite(a).all(), f'{what}: non-finite values'
    scale = max(b.abs().max().item(), 1e-12)
    err = (a - b).abs().max().item() / scale
    assert err < tol, f'{what}: rel-to-max err {err:.3e} >= {tol:.1e} (scale {scale:.3e})'


def q(t, dtype):
    """round a host tensor through the kernel's storage dtype (so the reference sees the same operands)"""
    return t.to(dtype).float()


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K', [(300, 168, 56), (1024, 256, 448), (640, 56, 224), (200, 84, 32), (128, 3584, 896), (16400, 392, 224), (4000, 2692, 448)])
def test_gemm_nt_epilogues(dtype, M, N, K):      # (16400, 392, 224): >= 384 tiles of 128 x 128 with K = 224 -> gemm_ring64_kernel; the last: 176 tiles of 256 x 256 with ragged edges -> gemm_big_kernel
    x, w = q(rnd(M, K), dtype), q(rnd(N, K, seed=1) * 0.1, dtype)
    bias, res = rnd(N, seed=2), rnd(M, N, seed=3)
    rows_per = 100
    rs = torch.rand((M + rows_per - 1) // rows_per) + 0.5
    xd, wd = x.to(DEV, dtype), w.to(DEV, dtype)
    y = ops().gemm(xd, wd, M, N, K, bias=bias.to(DEV))
    close(y, F.linear(x, w, bias), TOL[dtype], 'bias')
    y = ops().gemm(xd, wd, M, N, K, bias=bias.to(DEV), rowscale=rs.to(DEV), rows_per_scale=rows_per, residual=res.to(DEV),
                   out_dtype=torch.float32)
    ref = res + F.linear(x, w, bias) * rs.repeat_interleave(rows_per)[:M, None]
    close(y, ref, TOL[dtype], 'bias+rowscale+residual')
    y = ops().gemm(xd, wd, M, N, K, bias=bias.to(DEV), act=1, slope=0.01, out_dtype=torch.float32)
    close(y, F.leaky_relu(F.linear(x, w, bias), 0.01), TOL[dtype], 'lrelu')
    aux = q(rnd(M, N, seed=5), dtype)
    y = ops().gemm(xd, wd, M, N, K, act=2, aux=aux.to(DEV, dtype), out_dtype=torch.float32)
    a64 = aux.double()
    gp = 0.5 * (1 + torch.erf(a64 / math.sqrt(2))) + a64 * torch.exp(-0.5 * a64 * a64) / math.sqrt(2 * math.pi)
    close(y, F.linear(x, w).double() * gp, TOL[dtype], 'gelu-grad epilogue')
    g = torch.empty(M, N, device=DEV, dtype=dtype)
    y = ops().gemm(xd, wd, M, N, K, bias=bias.to(DEV), out_gelu=g)
    close(g, F.gelu(F.linear(x, w, bias)), TOL[dtype], 'second output GELU(v)')
    y = ops().gemm(xd, wd, M, N, K, x_op=1, out_dtype=torch.float32)
    close(y, F.linear(q(F.gelu(x), dtype) if dtype == torch.bfloat16 else F.gelu(x), w), TOL[dtype] * 2, 'gelu on load')


@pytest.mark.parametrize('dtype', DTYPES)
def test_gemm_c28_padded_rows(dtype):
    """C = 28 (encoder stage 0): rows padded to ld = 32, K*sizeof(T) not a multiple of 16 -> tail masked in registers;
    output written into a column slice that is only 8-byte aligned (bf16)."""
    M, K, N = 500, 28, 56
    x, w = q(rnd(M, K), dtype), q(rnd(N, K, seed=1) * 0.2, dtype)
    xp = torch.full((M, 32), float('nan')); xp[:, :K] = x
    wp = torch.full((N, 32), float('nan')); wp[:, :K] = w
    buf = torch.zeros(M, 88, device=DEV, dtype=dtype)
    ops().gemm(xp.to(DEV, dtype)[:, :K], wp.to(DEV, dtype)[:, :K], M, N, K, out=buf[:, 32:88])
    close(buf[:, 32:88], x @ w.t(), TOL[dtype], 'C=28 NT into a column slice')
    assert float(buf[:, :32].abs().max()) == 0.0
    # dX = dY[M,56] @ W[56,28] with dY a column slice, and dW = dY^T X
    dy = q(rnd(M, N, seed=2), dtype)
    buf[:, 32:88] = dy.to(DEV, dtype)
    w2 = q(rnd(N, K, seed=3) * 0.2, dtype)
    w2p = torch.full((N, 32), float('nan')); w2p[:, :K] = w2
    dx = ops().gemm(buf[:, 32:88], w2p.to(DEV, dtype)[:, :K], M, K, N, w_trans=True, out_dtype=torch.float32)
    close(dx, dy @ w2, TOL[dtype], 'C=28 NN')
    dw = torch.zeros(N, K, device=DEV)
    ops().gemm(buf[:, 32:88], xp.to(DEV, dtype)[:, :K], N, K, M, x_trans=True, w_trans=True, out=dw, accumulate=True)
    close(dw, dy.t() @ x, TOL[dtype], 'C=28 TN')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K', [(300, 56, 168), (1024, 448, 256), (512, 224, 56), (12300, 508, 192), (8200, 508, 224), (4000, 2692, 448)])
def test_gemm_nn_dx(dtype, M, N, K):
    """dX[M,N] = dY[M,K] @ W[K,N]  (W stored [K][N] -> w_trans).  The last shape (>= 384 tiles, K % 64 == 0) takes the bf16
    kernel that reads W with transposing LDS reads (gemm_tr_ring_kernel<false, 64, ..>), including its GELU' epilogue and bf16 output;
    the one after it (K = 224: a multiple of 32 only) the 32-deep form with X in 64-byte LDS rows (gemm_tr_ring_kernel<false, 32, 4>); the last
    (176 ragged tiles of 256 x 256) the 8-wave kernel with W as two token-major images (gemm_big_kernel<.., true, ..>)."""
    dy, w = q(rnd(M, K), dtype), q(rnd(K, N, seed=1) * 0.1, dtype)
    ldn = (N + 7) // 8 * 8
    wp = torch.full((K, ldn), float('nan')); wp[:, :N] = w
    wd = wp.to(DEV, dtype)[:, :N]
    y = ops().gemm(dy.to(DEV, dtype), wd, M, N, K, w_trans=True, out_dtype=torch.float32)
    close(y, dy @ w, TOL[dtype], 'NN')
    aux = q(rnd(M, N, seed=5), dtype)
    out = torch.full((M, ldn), 5.0, device=DEV, dtype=dtype)
    ops().gemm(dy.to(DEV, dtype), wd, M, N, K, w_trans=True, out=out[:, :N], act=2, aux=aux.to(DEV, dtype))
    a64 = aux.double()
    gp = 0.5 * (1 + torch.erf(a64 / math.sqrt(2))) + a64 * torch.exp(-0.5 * a64 * a64) / math.sqrt(2 * math.pi)
    close(out[:, :N], (dy.double() @ w.double()) * gp, TOL[dtype], 'NN gelu-grad epilogue')
    if ldn > N:
        assert float((out[:, N:].float() - 5.0).abs().max()) == 0.0


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K,split', [(168, 56, 1000, 1), (256, 448, 4096, 4), (56, 224, 8192, 8), (84, 28 + 4, 2048, 2)])
def test_gemm_tn_dw(dtype, M, N, K, split):
    """dW[M,N] = dY[K,M]^T @ X[K,N] with split-K atomics into a zeroed f32 buffer"""
    dy, x = q(rnd(K, M), dtype), q(rnd(K, N, seed=1), dtype)
    out = torch.zeros(M, N, device=DEV)
    pad = lambda t: torch.cat([t, torch.full((t.shape[0], (-t.shape[1]) % 8), float('nan'))], 1).to(DEV, dtype)[:, :t.shape[1]]
    xsum = torch.zeros(M, device=DEV)
    ops().gemm(pad(dy), pad(x), M, N, K, x_trans=True, w_trans=True, out=out, accumulate=True, splitk=split, xsum=xsum)
    close(out, dy.t() @ x, TOL[dtype], 'TN')
    close(xsum, dy.sum(0), 1e-4, 'fused column sums (bias gradient)')
    # the production path: slab of partial tiles + fw_slab_reduce, accumulating into pre-filled targets
    dw, db = torch.ones(M, N, device=DEV), torch.ones(M, device=DEV)
    ops().wgrad(pad(dy), pad(x), M, N, K, dw, db)
    close(dw - 1, dy.t() @ x, TOL[dtype], 'wgrad (slab split-K)')
    close(db - 1, dy.sum(0), 1e-4, 'wgrad bias')


def test_wgrad_group_one_launch_for_a_whole_backward_pass():
    """fwair.ops.wgrad(defer=True) inside a backward pass queues the product; the end-of-pass callback launches ALL of them at once
    (gemm_wgrad_group_kernel): un-split deep-stage shapes that add straight into the gradient (on top of what is already there),
    tall reductions cut into slices whose partial tiles go through the slab fold, bias gradients both ways, ragged tiles."""
    from fwair import ops as OP
    shapes = [(448, 1792, 4096), (1344, 448, 4096), (56, 224, 40000 - 40000 % 32), (28, 28, 98304), (200, 72, 8192 + 64), (896, 896, 1024),
              (224, 672, 16384), (300, 260, 8192 + 32)]       # outputs >= 224 on both sides take the 256 x 256 tile kernel (the last: ragged, sliced)
    data = []
    for i, (n, k, m) in enumerate(shapes):
        g = (rnd(m, n, seed=i) * 0.3).to(torch.bfloat16)
        x = (rnd(m, k, seed=100 + i) * 0.3).to(torch.bfloat16)
        dw0 = rnd(n, k, seed=200 + i)                          # the gradient buffer already holds something: the product ADDS
        data.append((g, x, dw0))

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t * 1.0

        @staticmethod
        def backward(ctx, d):
            for (g, x, dw0), (dwd, dbd, gd, xd) in zip(data, dev):
                OP.wgrad(gd, xd, gd.shape[1], xd.shape[1], gd.shape[0], dwd, dbd, defer=True)
            assert len(OP._pending_w) == len(data), 'the products must be queued, not launched one by one'
            return d

    def padded(t):                                             # rows 16-byte aligned (ld multiple of 8), as the model's activations are
        ld = (t.shape[1] + 7) // 8 * 8
        buf = torch.zeros(t.shape[0], ld, dtype=t.dtype, device=DEV)
        buf[:, :t.shape[1]] = t.to(DEV)
        return buf[:, :t.shape[1]]

    dev = [(dw0.clone().to(DEV), torch.zeros(g.shape[1], device=DEV), padded(g), padded(x)) for g, x, dw0 in data]
    t = torch.ones(1, device=DEV, requires_grad=True)
    Fn.apply(t).sum().backward()
    assert not OP._pending_w and not OP._pending
    for (g, x, dw0), (dwd, dbd, _, _) in zip(data, dev):
        ref = dw0.double() + g.double().t() @ x.double()
        close(dwd, ref, 2e-3, f'dW {tuple(ref.shape)} over {g.shape[0]} tokens')
        close(dbd, g.double().sum(0), 2e-3, 'db')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K', [(32768 + 45, 224, 56), (33000, 56, 28), (32800, 448, 112), (32768, 132, 224), (40000, 672, 224),
                                   (32790, 28, 112)])
def test_gemm_stream_tall_skinny(dtype, M, N, K):
    """M >= 32768 with K*sizeof(T) <= 512 takes gemm_stream_kernel (W panel in LDS, X fragments straight from HBM):
    NT and NN forms, ragged last strip, N not a multiple of 64, K tails, every epilogue, several column panels."""
    if dtype == torch.float32 and K > 128:
        K = 128 - 4                                         # f32: 512 bytes of K at most on this path
    x, w = q(rnd(M, K), dtype), q(rnd(N, K, seed=1) * 0.1, dtype)
    ldk, ldn = (K + 7) // 8 * 8, (N + 7) // 8 * 8
    xp = torch.full((M, ldk), float('nan')); xp[:, :K] = x
    wp = torch.full((N, ldk), float('nan')); wp[:, :K] = w
    xd, wd = xp.to(DEV, dtype)[:, :K], wp.to(DEV, dtype)[:, :K]
    bias, res = rnd(N, seed=2), rnd(M, N, seed=3)
    rows_per = 4096
    rs = torch.rand((M + rows_per - 1) // rows_per) + 0.5
    ref = F.linear(x.double(), w.double())
    y = ops().gemm(xd, wd, M, N, K, bias=bias.to(DEV))
    close(y, ref + bias.double(), TOL[dtype], 'stream NT bias')
    y = ops().gemm(xd, wd, M, N, K, bias=bias.to(DEV), rowscale=rs.to(DEV), rows_per_scale=rows_per, residual=res.to(DEV),
                   out_dtype=torch.float32)
    close(y, res.double() + (ref + bias.double()) * rs.double().repeat_interleave(rows_per)[:M, None], TOL[dtype], 'stream NT rowscale+residual')
    aux = q(rnd(M, N, seed=5), dtype)
    y = ops().gemm(xd, wd, M, N, K, act=2, aux=aux.to(DEV, dtype), out_dtype=torch.float32)
    a64 = aux.double()
    gp = 0.5 * (1 + torch.erf(a64 / math.sqrt(2))) + a64 * torch.exp(-0.5 * a64 * a64) / math.sqrt(2 * math.pi)
    close(y, ref * gp, TOL[dtype], 'stream NT gelu-grad epilogue')
    g = torch.full((M, ldn), 7.0, device=DEV, dtype=dtype)
    y = ops().gemm(xd, wd, M, N, K, bias=bias.to(DEV), out_gelu=g[:, :N])
    close(g[:, :N], F.gelu(ref + bias.double()), TOL[dtype], 'stream NT second output')
    if ldn > N:
        assert float((g[:, N:].float() - 7.0).abs().max()) == 0.0, 'padding columns were written'
    # NN: dX[M,N] = dY[M,K] W[K,N]
    w2 = q(rnd(K, N, seed=6) * 0.1, dtype)
    w2p = torch.full((K, ldn), float('nan')); w2p[:, :N] = w2
    y = ops().gemm(xd, w2p.to(DEV, dtype)[:, :N], M, N, K, w_trans=True, out_dtype=torch.float32)
    close(y, x.double() @ w2.double(), TOL[dtype], 'stream NN')


@pytest.mark.parametrize('dtype', DTYPES)
def test_dgrad_split_reduction(dtype):
    """Input gradient of the 65536-wide mlp_head (encoder_Uformer.py:942): small output, very long reduction -> split
    over N into partial tiles + slab reduce (ops.dgrad)."""
    M, K, N = 256, 448, 16384
    g, w = q(rnd(M, N), dtype), q(rnd(N, K, seed=1) * 0.05, dtype)
    out = torch.full((M, K + 8), 3.0, device=DEV, dtype=dtype)
    ops().dgrad(g.to(DEV, dtype), w.to(DEV, dtype), M, K, N, out[:, :K])
    close(out[:, :K], g.double() @ w.double(), TOL[dtype], 'split dgrad')
    assert float((out[:, K:].float() - 3.0).abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,C', [(1000, 28), (777, 56), (512, 112), (300, 448), (130, 896)])
def test_layernorm(dtype, rows, C):
    x, g, b = rnd(rows, C) * 2 + 0.3, 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2)
    y, mean, rstd = ops().layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV), dtype)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br)
    close(y, ref, TOL[dtype], 'y')
    dy = q(rnd(rows, C, seed=3), dtype)
    dres = rnd(rows, C, seed=4)
    ref.backward(dy)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops().layernorm_bwd(dy.to(DEV, dtype), x.to(DEV), g.to(DEV), mean, rstd, dg, db, dres=dres.to(DEV))
    close(dx, xr.grad + dres, 1e-4, 'dx')
    close(dg, gr.grad, 1e-4, 'dgamma')
    close(db, br.grad, 1e-4, 'dbeta')


# ------------------------------------------------------------------------------------------------ attention
def ref_window_attention(qkv, C, B, H, W, heads, L, mode, shift, tables, lam=None, nb=3):
    """CPU reference built from the oracle's pieces.  qkv: [L*B*H*W, 3C] f32/f64.  tables: [ntab,225,heads].
    Decoder semantics when L == 1 (lam: [B, nb-1, heads] or None); encoder intra/inter otherwise."""
    D = C // heads
    N = 64
    nW = (H // 8) * (W // 8)
    x = qkv.view(L * B, H, W, 3 * C)
    if shift:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    xw = O.window_partition(x, 8).view(L, B * nW, N, 3, heads, D)           # l, bnw, tok, qkv, h, d
    qh, kh, vh = (xw[:, :, :, i].permute(1, 3, 0, 2, 4) for i in range(3))  # bnw, h, l, tok, d
    mask = O.shift_attn_mask(H, W, 8, shift, qkv.dtype) if shift else None
    outs = []
    for lq in range(L):
        keys = [lq] if mode == 0 else [l for l in range(L) if l != lq]
        s = torch.cat([(qh[:, :, lq] * D ** -0.5) @ kh[:, :, lk].transpose(-1, -2)
                       + O.rel_bias(tables[lq * L + lk], 8).unsqueeze(0)
                       + (mask.repeat(B, 1, 1).unsqueeze(1) if mask is not None else 0) for lk in keys], -1)
        p = s.softmax(-1)
        if lam is not None:
            kind, size = ('frequency_decompose_1', 1. / (nb - 1)) if nb == 3 else ('frequency_decompose_dc', 0.5)
            bands = O.frequency_decompose(p, kind, size, N, N, True)
            for i in range(1, bands.shape[0]):
                p = p + (bands[i].view(B, nW, heads, N, N) * lam[:, i - 1][:, None, :, None, None]).view(-1, heads, N, N)
        outs.append(p @ torch.cat([vh[:, :, lk] for lk in keys], -2))        # bnw, h, tok, d
    o = torch.stack(outs, 0).permute(0, 1, 3, 2, 4).reshape(L * B * nW, 8, 8, C)
    o = O.window_reverse(o, 8, H, W)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    return o.reshape(L * B * H * W, C)


def lam_to_coef(lam, nb):
    if nb == 3:
        l1, l2 = lam[:, 0], lam[:, 1]
        return torch.stack([1 + l2, -l2 / 64, l1 - l2], -1)
    l1 = lam[:, 0]
    return torch.stack([1 + l1, -l1 / 64, torch.zeros_like(l1)], -1)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('heads,shift,lfs', [(2, 0, 2), (2, 4, 2), (1, 4, 1), (2, 4, 0), (4, 0, 2)])
def test_attention_decoder(dtype, heads, shift, lfs):
    B, H, W, C = 2, 16, 16, 56 * heads
    nb = {0: 0, 1: 2, 2: 3}[lfs]
    qkv = q(rnd(B * H * W, 3 * C), dtype).requires_grad_(True)
    tables = (rnd(1, 225, heads, seed=1) * 0.5).requires_grad_(True)
    lam = (rnd(B, max(nb - 1, 1), heads, seed=2) * 0.3).requires_grad_(True) if lfs else None
    ref = ref_window_attention(qkv, C, B, H, W, heads, 1, 0, shift, tables, lam, nb)
    coef = lam_to_coef(lam, nb).detach().to(DEV).contiguous() if lfs else None
    qd = qkv.detach().to(DEV, dtype)
    out, lse = ops().attn_fwd(qd, C, B, H, W, heads, 1, 0, shift, tables.detach().to(DEV), coef, lfs)
    close(out, ref, TOL[dtype], 'out')
    dout = q(rnd(B * H * W, C, seed=3), dtype)
    ref.backward(dout)
    lam_grad_ref = lam.grad.clone() if lfs else None
    dbias = torch.zeros(1, 225, heads, device=DEV)
    dcoef = torch.zeros(B, heads, 3, device=DEV) if lfs else None
    dqkv = ops().attn_bwd(qd, out, dout.to(DEV, dtype), lse, C, B, H, W, heads, 1, 0, shift, tables.detach().to(DEV), dbias,
                          coef, dcoef, lfs)
    tol = TOL[dtype] * (3 if dtype == torch.bfloat16 else 4)
    close(dqkv, qkv.grad, tol, 'dqkv')
    close(dbias, tables.grad, tol, 'dbias table')
    if lfs:
        lam.grad = None
        (lam_to_coef(lam, nb) * dcoef.cpu()).sum().backward()     # chain rule coef -> lambda on the host
        close(lam.grad, lam_grad_ref, tol, 'dlambda')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('L,mode,shift', [(3, 0, 4), (3, 1, 0), (3, 1, 4), (2, 1, 4), (1, 0, 4)])
def test_attention_encoder(dtype, L, mode, shift):
    B, H, W, heads = 2, 16, 16, 2
    C = 28 * heads
    qkv = q(rnd(L * B * H * W, 3 * C), dtype).requires_grad_(True)
    tables = (rnd(L * L, 225, heads, seed=1) * 0.5).requires_grad_(True)
    ref = ref_window_attention(qkv, C, B, H, W, heads, L, mode, shift, tables)
    # activations with C = 56 have ld = 168: fine (multiple of 8)
    qd = qkv.detach().to(DEV, dtype)
    out, lse = ops().attn_fwd(qd, C, B, H, W, heads, L, mode, shift, tables.detach().to(DEV))
    close(out, ref, TOL[dtype], 'out')
    dout = q(rnd(L * B * H * W, C, seed=3), dtype)
    ref.backward(dout)
    dbias = torch.zeros(L * L, 225, heads, device=DEV)
    dqkv = ops().attn_bwd(qd, out, dout.to(DEV, dtype), lse, C, B, H, W, heads, L, mode, shift, tables.detach().to(DEV), dbias)
    tol = TOL[dtype] * 4
    close(dqkv, qkv.grad, tol, 'dqkv')
    close(dbias, tables.grad, tol, 'dbias tables')


@pytest.mark.parametrize('rows,cols,ld', [(70000, 28, 64), (262144, 28, 56), (4099, 224, 224), (513, 512, 512), (1000, 516, 520), (300, 30, 32)])
def test_colsum(rows, cols, ld):
    """fw_colsum (bias gradients): the float4 form for narrow matrices (cols % 4 == 0, <= 512) and the plain form otherwise; ADDS into out."""
    x = rnd(rows, ld, seed=7)[:, :cols]
    out0 = rnd(cols, seed=8)
    out = out0.clone().to(DEV)
    ops().colsum(x.to(DEV)[:, :cols] if ld == cols else x.to(DEV), out)
    close(out, out0.double() + x.double().sum(0), 2e-5, f'colsum {rows}x{cols}')


# ------------------------------------------------------------------------------------------------ LeFF dwconv
@pytest.mark.parametrize('twin', [True, False])
@pytest.mark.parametrize('dtype', DTYPES)
def test_dwconv(dtype, twin):
    """twin: the caller kept g1 = GELU(h1); else (the model's path) the kernels take the pre-activation h1 and apply GELU on load --
    the forward stencil and the input-centric weight gradient."""
    B, H, W, C = 2, 16, 16, 112
    h1 = q(rnd(B * H * W, C), dtype).requires_grad_(True)
    w = (rnd(C, 1, 3, 3, seed=1) * 0.3).requires_grad_(True)
    b = (rnd(C, seed=2) * 0.1).requires_grad_(True)
    g1 = q(F.gelu(h1.detach()), dtype)                       # the stored post-activation twin (rounded like the kernel stores it)
    ref = F.conv2d(F.gelu(h1).view(B, H, W, C).permute(0, 3, 1, 2), w, b, padding=1, groups=C).permute(0, 2, 3, 1).reshape(B * H * W, C)
    wt = w.detach().view(C, 9).t().contiguous().to(DEV)                 # tap-major [9, C]
    if twin:
        h2, g2 = ops().dwconv_fwd(g1.to(DEV, dtype), wt, b.detach().to(DEV), B, H, W)
    else:
        h2, g2 = ops().dwconv_fwd(h1.detach().to(DEV, dtype), wt, b.detach().to(DEV), B, H, W, in_gelu=True)
    close(h2, ref, TOL[dtype], 'h2')
    close(g2, F.gelu(ref), TOL[dtype], 'g2')
    dh2 = q(rnd(B * H * W, C, seed=3), dtype)
    ref.backward(dh2)
    dw, db = torch.zeros(C, 9, device=DEV), torch.zeros(C, device=DEV)
    dh1 = ops().dwconv_bwd(dh2.to(DEV, dtype), g1.to(DEV, dtype) if twin else None, h1.detach().to(DEV, dtype), wt, dw, db, B, H, W)
    close(dh1, h1.grad, TOL[dtype] * 2, 'dh1')
    close(dw, w.grad.view(C, 9), TOL[dtype] * 2, 'dw')
    close(db, b.grad, TOL[dtype] * 2, 'db')


# ------------------------------------------------------------------------------------------------ convs as GEMM
@pytest.mark.parametrize('dtype', DTYPES)
def test_downsample_conv(dtype):
    B, H, W, C = 2, 16, 16, 56
    x = rnd(B * H * W, C).requires_grad_(True)
    w = (rnd(2 * C, C, 4, 4, seed=1) * 0.05).requires_grad_(True)
    b = rnd(2 * C, seed=2)
    ref = F.conv2d(x.view(B, H, W, C).permute(0, 3, 1, 2), w, b, stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, 2 * C)
    wk = w.detach().permute(0, 2, 3, 1).reshape(2 * C, 16 * C).contiguous()          # [Cout][(ky,kx,ci)]
    wq = torch.empty(2 * C, 16 * C, device=DEV, dtype=dtype)
    ops().permute3(w.detach().to(DEV).contiguous(), wq, (2 * C, C, 16), (16 * C, 1, C))
    close(wq, wk, TOL[dtype] if dtype == torch.bfloat16 else 1e-7, 'weight re-layout')
    col = ops().im2col4(x.detach().to(DEV), B, H, W, dtype)
    y = ops().gemm(col, wq, B * (H // 2) * (W // 2), 2 * C, 16 * C, bias=b.to(DEV), out_dtype=torch.float32)
    close(y, ref, TOL[dtype], 'conv k4s2p1')
    dy = rnd(B * (H // 2) * (W // 2), 2 * C, seed=3)
    ref.backward(dy)
    dyq = ops().cast_rows(dy.to(DEV), dtype)
    dcol = ops().gemm(dyq, wq, dyq.shape[0], 16 * C, 2 * C, w_trans=True)
    dres = rnd(B * H * W, C, seed=4)
    dx = ops().col2im4(dcol, B, H, W, C, dres=dres.to(DEV))
    close(dx, x.grad + dres, TOL[dtype] * 2, 'dx')
    dwk = torch.zeros(2 * C, 16 * C, device=DEV)
    ops().gemm(dyq, col, 2 * C, 16 * C, dyq.shape[0], x_trans=True, w_trans=True, out=dwk, accumulate=True, splitk=2)
    dw = torch.zeros(2 * C, C, 16, device=DEV)
    ops().permute3(dwk, dw, (2 * C, 16, C), (16 * C, 1, 16), accumulate=True)
    close(dw.view_as(w), w.grad, TOL[dtype] * 2, 'dw')


@pytest.mark.parametrize('dtype', DTYPES)
def test_upsample_convT(dtype):
    B, H, W, Cin, Cout = 2, 8, 8, 112, 56
    x = q(rnd(B * H * W, Cin), dtype).requires_grad_(True)
    w = (rnd(Cin, Cout, 2, 2, seed=1) * 0.1).requires_grad_(True)
    b = rnd(Cout, seed=2)
    ref = F.conv_transpose2d(x.view(B, H, W, Cin).permute(0, 3, 1, 2), w, b, stride=2).permute(0, 2, 3, 1).reshape(-1, Cout)
    wt = torch.empty(4 * Cout, Cin, device=DEV, dtype=dtype)                          # [(i,j,co)][ci]
    ops().permute3(w.detach().to(DEV).contiguous(), wt, (Cin, Cout, 4), (1, Cin, Cout * Cin))
    g = ops().gemm(x.detach().to(DEV, dtype), wt, B * H * W, 4 * Cout, Cin)
    out = torch.zeros(B * 4 * H * W, 2 * Cout, device=DEV)                            # left half of a concat buffer
    ops().pixel_shuffle(g, b.to(DEV), out[:, :Cout], B, H, W, Cout)
    close(out[:, :Cout], ref, TOL[dtype], 'convT k2s2')
    assert float(out[:, Cout:].abs().max()) == 0.0
    dy = rnd(B * 4 * H * W, Cout, seed=3)
    ref.backward(dy)
    dcat = torch.zeros(B * 4 * H * W, 2 * Cout, device=DEV)
    dcat[:, :Cout] = dy.to(DEV)
    dg = ops().pixel_unshuffle(dcat[:, :Cout], B, H, W, Cout, dtype)
    dx = ops().gemm(dg, wt, B * H * W, Cin, 4 * Cout, w_trans=True, out_dtype=torch.float32)
    close(dx, x.grad, TOL[dtype] * 2, 'dx')
    dwt = torch.zeros(4 * Cout, Cin, device=DEV)
    ops().gemm(dg, x.detach().to(DEV, dtype), 4 * Cout, Cin, B * H * W, x_trans=True, w_trans=True, out=dwt, accumulate=True)
    dw = torch.zeros(Cin, Cout, 4, device=DEV)
    ops().permute3(dwt, dw, (4, Cout, Cin), (1, 4, Cout * 4), accumulate=True)
    close(dw.view_as(w), w.grad, TOL[dtype] * 2, 'dw')
    db = torch.zeros(Cout, device=DEV)
    ops().colsum(dcat[:, :Cout], db)
    close(db, dy.sum(0), 1e-4, 'dbias')


def test_in_out_proj():
    B, H, W, C = 2, 16, 16, 56
    img = rnd(B, 3, H, W)
    w = (rnd(C, 3, 3, 3, seed=1) * 0.2).requires_grad_(True)
    b = (rnd(C, seed=2) * 0.1).requires_grad_(True)
    ref = F.leaky_relu(F.conv2d(img, w, b, padding=1), 0.01).permute(0, 2, 3, 1).reshape(-1, C)
    out = torch.empty(B * H * W, C, device=DEV)
    call('fw_inproj_fwd', img.to(DEV), w.detach().to(DEV), b.detach().to(DEV), out, C, B, H, W, C, 0.01)
    close(out, ref, 2e-5, 'inproj')
    dy = rnd(B * H * W, C, seed=3)
    ref.backward(dy)
    dw, db = torch.zeros_like(w, device=DEV), torch.zeros(C, device=DEV)
    call('fw_inproj_bwd', img.to(DEV), out, C, dy.to(DEV), C, dw, db, B, H, W, C, 0.01)
    close(dw, w.grad, 1e-4, 'inproj dw')
    close(db, b.grad, 1e-4, 'inproj db')
    # output projection + global residual
    C2 = 112
    fea = rnd(B * H * W, C2, seed=4).requires_grad_(True)
    w2 = (rnd(3, C2, 3, 3, seed=5) * 0.1).requires_grad_(True)
    b2 = (rnd(3, seed=6) * 0.1).requires_grad_(True)
    ref = img + F.conv2d(fea.view(B, H, W, C2).permute(0, 3, 1, 2), w2, b2, padding=1)
    o = torch.empty(B, 3, H, W, device=DEV)
    call('fw_outproj_fwd', fea.detach().to(DEV), C2, w2.detach().to(DEV), b2.detach().to(DEV), img.to(DEV), o, B, H, W, C2)
    close(o, ref, 2e-5, 'outproj')
    do = rnd(B, 3, H, W, seed=7)
    ref.backward(do)
    dfea = torch.empty(B * H * W, C2, device=DEV)
    dw2, db2 = torch.zeros_like(w2, device=DEV), torch.zeros(3, device=DEV)
    call('fw_outproj_bwd', do.to(DEV), fea.detach().to(DEV), C2, w2.detach().to(DEV), dfea, C2, dw2, db2, B, H, W, C2)
    close(dfea, fea.grad, 1e-4, 'outproj dfea')
    close(dw2, w2.grad, 1e-4, 'outproj dw')
    close(db2, b2.grad, 1e-4, 'outproj db')


# ------------------------------------------------------------------------------------------------ losses / optimiser
def test_losses_adam_ema():
    a, b = rnd(2, 3, 32, 32).requires_grad_(True), rnd(2, 3, 32, 32, seed=1)
    loss = torch.zeros(1, device=DEV)
    da = torch.empty_like(a, device=DEV)
    call('fw_l1_loss', a.detach().to(DEV), b.to(DEV), da, a.numel(), 0.7, loss)
    ref = F.l1_loss(a, b)
    (0.7 * ref).backward()
    close(loss, ref.reshape(1), 1e-5, 'l1')
    close(da, a.grad, 1e-5, 'l1 grad')
    lg = (rnd(6, 49, seed=2) * 3).requires_grad_(True)
    loss.zero_()
    dl = torch.empty(6, 49, device=DEV)
    call('fw_ce0_loss', lg.detach().to(DEV), dl, 6, 49, 0.6, loss)
    ref = F.cross_entropy(lg, torch.zeros(6, dtype=torch.long))
    (0.6 * ref).backward()
    close(loss, ref.reshape(1), 1e-5, 'ce')
    close(dl, lg.grad, 1e-5, 'ce grad')
    # Adam, 3 steps, against torch.optim.Adam
    n = 10007
    p0 = rnd(n)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=2e-4)
    pd, m, v = p0.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sh = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    hyper = torch.tensor([2e-4, 1.0, 1.0, 0.0], device=DEV)
    for s in range(3):
        g = rnd(n, seed=10 + s)
        p.grad = g.clone()
        opt.step()
        call('fw_adam_tick', hyper, 0.9, 0.999)
        call('fw_adam', 1, pd, g.to(DEV), m, v, sh, n, hyper, 0.9, 0.999, 1e-8)
    close(pd, p.detach(), 1e-6, 'adam params')
    close(sh.float(), p.detach(), 1e-2, 'adam bf16 shadow')
    pk, pq = rnd(n, seed=20), rnd(n, seed=21)
    pkd = pk.to(DEV).clone()
    call('fw_ema', 0, pkd, pq.to(DEV), None, n, 0.999)
    close(pkd, pk * 0.999 + pq * (1 - 0.999), 1e-6, 'ema')


# ------------------------------------------------------------------------------------------------ band decomposition
@pytest.mark.parametrize('N', [64, 128, 256])
def test_dft_band_decomposition(N):
    from fwair import lfs
    x = rnd(3, N, N)
    fr, fi = torch.empty(3, N, N, device=DEV), torch.empty(3, N, N, device=DEV)
    call('fw_dft2_fwd', x.to(DEV), fr, fi, 3, N)
    F2 = torch.fft.fft2(x.double())
    close(fr, F2.real, 2e-5, 'spectrum re')
    close(fi, F2.imag, 2e-5, 'spectrum im')
    for kind, size in (('frequency_decompose_1', 0.5), ('frequency_decompose', 1 / 3.)):
        masks = torch.stack(lfs.band_masks_shifted(kind, size, N, N)).float()
        mu = torch.fft.ifftshift(masks, dim=(-2, -1)).contiguous().to(DEV)
        nb = masks.shape[0]
        out = torch.empty(nb, 3, N, N, device=DEV)
        call('fw_dft2_bands', fr, fi, mu, out, 3, N, nb, 0)
        ref = O.frequency_decompose(x.view(1, 3, N, N), kind, size, N, N, True)[:, 0]
        close(out, ref, 2e-5, f'{kind} real bands')
        assert bool((masks.sum(0) == 1).all())                      # the masks partition the spectrum ...
        outp = torch.full((nb, 3, N, N), float('nan'), device=DEV)  # ... so the last band is x minus the others
        call('fw_dft2_bands', fr, fi, mu, outp, 3, N, nb - 1, 0)
        call('fw_band_residual', x.to(DEV), outp, 3, N, nb)
        close(outp, ref, 2e-5, f'{kind} real bands, last by subtraction')
        if N in (64, 128):                                          # the one-launch f32-MFMA form the model's encoder pre-processing takes
            from net.utils.frequency_decompose import FrequencyDecompose, _dft_panels
            dc = sum(1 << b for b in range(nb) if float(masks[b].sum()) == 1.0 and float(mu[b, 0, 0]) == 1.0)
            assert dc == (1 if kind == 'frequency_decompose_1' else 0)
            for bits in (dc, 0):                                     # with the DC band as the mean, and through the full transform
                outm = torch.full((nb, 3, N, N), float('nan'), device=DEV)
                call('fw_dft2_decompose', x.to(DEV), mu, _dft_panels(N, torch.device(DEV)), outm, 3, N, nb, bits)
                close(outm, ref, 2e-5, f'{kind} one-launch MFMA decomposition (dc_bits {bits})')
            mod = FrequencyDecompose(kind, size, N, N)
            close(mod(x.view(1, 3, N, N).to(DEV))[:, 0], ref, 2e-5, f'{kind} through the module')
        out2 = torch.empty(nb, 3, N, N, 2, device=DEV)
        call('fw_dft2_bands', fr, fi, mu, out2, 3, N, nb, 1)
        ref2 = O.frequency_decompose(x.view(1, 3, N, N).double(), kind, size, N, N, False)[:, 0]
        close(out2, ref2, 2e-5, f'{kind} spectrum')
    o = torch.empty(2, 3, N, N, device=DEV)
    call('fw_dc_split', x.to(DEV), o, 3, N * N)
    close(o, O.frequency_decompose(x.view(1, 3, N, N), 'frequency_decompose_dc', 0.5, N, N)[:, 0], 2e-5, 'dc split')


@pytest.mark.parametrize('inverse', [True, False])
@pytest.mark.parametrize('kind,size', [('frequency_decompose', 1 / 3.), ('frequency_decompose_1', 0.5), ('frequency_decompose_dc', 0.5)])
def test_frequency_decompose_backward(kind, size, inverse):
    """FrequencyDecompose as a differentiable module (the frequency L1 loss of train.py:69-70,90-91): forward and input gradient
    against the oracle's torch.fft restatement."""
    from net.utils.frequency_decompose import FrequencyDecompose
    N = 64
    x = rnd(2, 3, N, N)
    xo = x.clone().double().requires_grad_(True)
    ref = O.frequency_decompose(xo, kind, size, N, N, inverse)
    wgt = rnd(*ref.shape, seed=9).double()
    (ref * wgt).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    out = FrequencyDecompose(kind, size, N, N, inverse=inverse)(xd)
    close(out, ref, 2e-5, 'forward')
    (out * wgt.to(DEV).float()).sum().backward()
    close(xd.grad, xo.grad, 5e-5, 'input gradient')


# ------------------------------------------------------------------------------------------------ encoder head
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('P', [16384, 1000])          # whole 16-byte strides (the vector form of the statistic passes) / the scalar form
def test_bn_lrelu_gap(dtype, P):
    B, ED = 4, 8
    fea = q(rnd(B, ED, P) * 1.5 + 0.2, dtype).requires_grad_(True)
    g, b = (1 + 0.1 * rnd(ED, seed=1)).requires_grad_(True), (0.1 * rnd(ED, seed=2)).requires_grad_(True)
    rm, rv = rnd(ED, seed=3) * 0.1, 1 + rnd(ED, seed=4).abs() * 0.1
    bn_rm, bn_rv = rm.clone(), rv.clone()
    ref = F.leaky_relu(F.batch_norm(fea.view(B, ED, P, 1), bn_rm, bn_rv, g, b, True, 0.1, 1e-5), 0.1).mean((2, 3))
    rmd, rvd = rm.to(DEV).clone(), rv.to(DEV).clone()
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    part, saved, gap = torch.empty(ED, B, 2, device=DEV), torch.empty(ED, 2, device=DEV), torch.empty(B, ED, device=DEV)
    fd = fea.detach().to(DEV, dtype)
    call('fw_bn_lrelu_gap_fwd', 1 if dtype == torch.bfloat16 else 0, fd, g.detach().to(DEV), b.detach().to(DEV), rmd, rvd, nbt,
         part, saved, gap, B, ED, P, 1, 1e-5, 0.1, 0.1)
    close(gap, ref, 2e-4, 'gap')
    close(rmd, bn_rm, 1e-4, 'running_mean')
    close(rvd, bn_rv, 1e-4, 'running_var')
    assert int(nbt) == 1
    dgap = rnd(B, ED, seed=5)
    ref.backward(dgap)
    part2 = torch.empty(ED, B, 2, device=DEV)
    dfea = torch.empty(B, ED, P, device=DEV, dtype=dtype)
    dg, db = torch.zeros(ED, device=DEV), torch.zeros(ED, device=DEV)
    call('fw_bn_lrelu_gap_bwd', 1 if dtype == torch.bfloat16 else 0, fd, g.detach().to(DEV), b.detach().to(DEV), saved,
         dgap.to(DEV), part2, dfea, dg, db, B, ED, P, 0.1)
    close(dfea, fea.grad, TOL[dtype] * 4, 'dfea')
    close(dg, g.grad, 5e-3 if dtype == torch.bfloat16 else 2e-4, 'dgamma')
    close(db, b.grad, 5e-3 if dtype == torch.bfloat16 else 2e-4, 'dbeta')
    # eval mode uses running statistics
    gap2 = torch.empty(B, ED, device=DEV)
    call('fw_bn_lrelu_gap_fwd', 1 if dtype == torch.bfloat16 else 0, fd, g.detach().to(DEV), b.detach().to(DEV), rmd, rvd, nbt,
         None, None, gap2, B, ED, P, 0, 1e-5, 0.1, 0.1)
    ref2 = F.leaky_relu(F.batch_norm(fea.detach().view(B, ED, P, 1), bn_rm, bn_rv, g.detach(), b.detach(), False, 0.1, 1e-5), 0.1).mean((2, 3))
    close(gap2, ref2, 2e-4, 'gap eval')


# ------------------------------------------------------------------------------------------------ MoCo
def test_moco_logits():
    L, B, ED, K = 3, 4, 256, 12
    qv, kv = rnd(L, B, ED).requires_grad_(True), rnd(L, B, ED, seed=1)
    queue = F.normalize(rnd(L, ED, K, seed=2), dim=1)
    qn, kn = F.normalize(qv, dim=2), F.normalize(kv, dim=2)
    ref = torch.stack([torch.cat([(qn[i] * kn[i]).sum(1, keepdim=True), qn[i] @ queue[i]], 1) / 0.07 for i in range(L)])
    logits, khat = torch.empty(L, B, 1 + K, device=DEV), torch.empty(L, B, ED, device=DEV)
    call('fw_moco_logits', qv.detach().to(DEV), kv.to(DEV), queue.to(DEV), logits, khat, L, B, ED, K, 1 / 0.07)
    close(logits, ref, 2e-5, 'logits')
    close(khat, kn, 2e-5, 'khat')
    dl = rnd(L, B, 1 + K, seed=3)
    ref.backward(dl)
    dq = torch.empty(L, B, ED, device=DEV)
    call('fw_moco_logits_bwd', qv.detach().to(DEV), khat, queue.to(DEV), dl.to(DEV), dq, L, B, ED, K, 1 / 0.07)
    close(dq, qv.grad, 1e-4, 'dq')
    qd = queue.to(DEV).clone()
    ptr = torch.tensor([8], dtype=torch.long, device=DEV)
    call('fw_moco_enqueue', qd, khat, ptr, L, B, ED, K)
    qref = queue.clone()
    for i in range(L):
        qref[i][:, 8:12] = kn[i].t()
    close(qd, qref, 2e-5, 'enqueue')
    assert int(ptr) == 0


# ------------------------------------------------------------------------------------------------ LFS lambda heads
@pytest.mark.parametrize('nb,NT', [(3, 64), (2, 64), (3, 256)])      # NT = (S/16)^2 encoder tokens: 64 at 128x128, 256 at 256x256
def test_lfs_lambda_heads(nb, NT):
    B, C, heads_list = 3, 448, [1, 2, 16]
    nb1 = nb - 1
    inter = rnd(nb1 * B, NT, C).requires_grad_(True)
    names = ['mlp_head.%d.0.weight', 'mlp_head.%d.0.bias', 'mlp_head.%d.1.weight', 'mlp_head.%d.1.bias',
             'mlp.%d.0.weight', 'mlp.%d.0.bias', 'mlp.%d.2.weight', 'mlp.%d.2.bias']
    states, dev_p, dev_g = [], [], []
    for bi, h in enumerate(heads_list):
        st = {}
        for band in range(1, nb):
            shapes = [(C,), (C,), (h, C), (h,), (h, h), (h,), (h, h), (h,)]
            for n, shp in zip(names, shapes):
                t = rnd(*shp, seed=bi * 100 + band * 10 + len(st)) * (0.3 if len(shp) > 1 else 0.2)
                if n.endswith('0.weight') and len(shp) == 1:
                    t = t + 1
                st[n % band] = t.requires_grad_(True)
        states.append(st)
    # reference: the oracle's per-block lambda head
    lam_ref = []
    for st, h in zip(states, heads_list):
        lam_ref.append(torch.stack([O.lfs_lambda(st, '', i, inter.view(nb1, B, NT, C)[i - 1])[:, 0] for i in range(1, nb)], 1))
    xbar, stats = torch.empty(nb1 * B, C, device=DEV), torch.empty(nb1 * B, NT, 2, device=DEV)
    idev = inter.detach().to(DEV)
    call('fw_lfs_xbar', idev, xbar, stats, nb1, B, NT, C, 1e-5)
    ptab, gtab, keep = [], [], []
    for st, h in zip(states, heads_list):
        for band in (1, 2):
            for n in names:
                if band < nb:
                    p = st[n % band].detach().to(DEV).contiguous()
                    g = torch.zeros_like(p)
                    keep.append((st[n % band], p, g))
                    ptab.append(p.data_ptr())
                    gtab.append(g.data_ptr())
                else:
                    ptab.append(0)
                    gtab.append(0)
    ptab = torch.tensor(ptab, dtype=torch.int64, device=DEV)
    gtab = torch.tensor(gtab, dtype=torch.int64, device=DEV)
    heads = torch.tensor(heads_list, dtype=torch.int32, device=DEV)
    offs = np.cumsum([0] + [B * h * 3 for h in heads_list])
    coef_off = torch.tensor(offs[:-1], dtype=torch.int64, device=DEV)
    coef = torch.zeros(int(offs[-1]), device=DEV)
    save = torch.zeros(len(heads_list), 2, B, 16, 3, device=DEV)
    call('fw_lfs_lambda', xbar, ptab, heads, coef_off, coef, save, len(heads_list), B, C, nb1)
    dcoef_host = []
    for bi, h in enumerate(heads_list):
        c = coef[offs[bi]:offs[bi + 1]].view(B, h, 3).cpu()
        close(c, lam_to_coef(lam_ref[bi], nb), 5e-5, f'coef block {bi}')
        dcoef_host.append(rnd(B, h, 3, seed=50 + bi))
    loss = sum((lam_to_coef(l, nb) * d).sum() for l, d in zip(lam_ref, dcoef_host))
    loss.backward()
    dcoef = torch.cat([d.reshape(-1) for d in dcoef_host]).to(DEV)
    dxbar = torch.zeros(nb1 * B, C, device=DEV)
    call('fw_lfs_lambda_bwd', xbar, ptab, gtab, heads, coef_off, dcoef, save, dxbar, len(heads_list), B, C, nb1)
    for ref_p, p, g in keep:
        close(g, ref_p.grad, 2e-4, 'lambda-head parameter grad')
    dinter = torch.zeros(nb1 * B, NT, C, device=DEV)
    call('fw_lfs_xbar_bwd', idev, stats, dxbar, dinter, nb1, B, NT, C)
    close(dinter, inter.grad, 2e-4, 'dinter')


# ------------------------------------------------------------------------------------------------ shapes of the TIMED step (B = 16)
# The bench runs B = 16 at 128x128.  At that size three code paths fire that the small cases above never reach:
#   * fw_attn_fwd caps its grid at 4 096 workgroups and LOOPS over items (with a trailing barrier between items),
#   * fw_attn_bwd's chunks walk many windows per workgroup, accumulating the bias gradient across them,
#   * fw_dwconv_fwd / _bwd switch to the LDS-tiled kernel (dwconv_tile_kernel) once B*H*W*C >= 80 M elements.
@pytest.mark.parametrize('dtype', DTYPES)
def test_attention_decoder_timed_shape(dtype):
    """decoderlayer_0 of the timed step: B = 16, 128x128, C = 112, 2 heads, LFS (3 bands), shifted -> 8 192 items > the 4 096 cap."""
    B, H, W, heads, shift, lfs, nb = 16, 128, 128, 2, 4, 2, 3
    C = 56 * heads
    qkv = q(rnd(B * H * W, 3 * C), dtype).requires_grad_(True)
    tables = (rnd(1, 225, heads, seed=1) * 0.5).requires_grad_(True)
    lam = (rnd(B, nb - 1, heads, seed=2) * 0.3).requires_grad_(True)
    ref = ref_window_attention(qkv, C, B, H, W, heads, 1, 0, shift, tables, lam, nb)
    coef = lam_to_coef(lam, nb).detach().to(DEV).contiguous()
    qd = qkv.detach().to(DEV, dtype)
    out, lse = ops().attn_fwd(qd, C, B, H, W, heads, 1, 0, shift, tables.detach().to(DEV), coef, lfs)
    close(out, ref, TOL[dtype], 'out (8192 items)')
    dout = q(rnd(B * H * W, C, seed=3), dtype)
    ref.backward(dout)
    lam_grad_ref = lam.grad.clone()
    dbias = torch.zeros(1, 225, heads, device=DEV)
    dcoef = torch.zeros(B, heads, 3, device=DEV)
    dqkv = ops().attn_bwd(qd, out, dout.to(DEV, dtype), lse, C, B, H, W, heads, 1, 0, shift, tables.detach().to(DEV), dbias, coef, dcoef, lfs)
    tol = TOL[dtype] * (3 if dtype == torch.bfloat16 else 4)
    close(dqkv, qkv.grad, tol, 'dqkv (8192 items)')
    close(dbias, tables.grad, tol * 2, 'dbias table summed over 4096 windows')
    lam.grad = None
    (lam_to_coef(lam, nb) * dcoef.cpu()).sum().backward()
    close(lam.grad, lam_grad_ref, tol * 2, 'dlambda summed over 256 windows per image')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('mode', [0, 1])
def test_attention_encoder_timed_shape(dtype, mode):
    """encoderlayer_0 of the query encoder in the timed step: L = 3 bands x B = 16 images, 128x128, C = 28, 1 head, shifted:
    12 288 items (intra) / 12 288 items with two key tiles (inter)."""
    L, B, H, W, heads, shift = 3, 16, 128, 128, 1, 4
    C = 28 * heads
    qkv = q(rnd(L * B * H * W, 3 * C), dtype)
    ld = (3 * C + 7) // 8 * 8
    tables = (rnd(L * L, 225, heads, seed=1) * 0.5).requires_grad_(True)
    qr = qkv.clone().requires_grad_(True)
    ref = ref_window_attention(qr, C, B, H, W, heads, L, mode, shift, tables)
    buf = torch.zeros(L * B * H * W, ld, device=DEV, dtype=dtype)
    buf[:, :3 * C] = qkv.to(DEV, dtype)
    qd = buf[:, :3 * C]
    out, lse = ops().attn_fwd(qd, C, B, H, W, heads, L, mode, shift, tables.detach().to(DEV))
    close(out, ref, TOL[dtype], 'out (12288 items)')
    dout = q(rnd(L * B * H * W, C, seed=3), dtype)
    ref.backward(dout)
    dbias = torch.zeros(L * L, 225, heads, device=DEV)
    dbuf = torch.zeros(L * B * H * W, (C + 7) // 8 * 8, device=DEV, dtype=dtype)
    dbuf[:, :C] = dout.to(DEV, dtype)
    dqkv = ops().attn_bwd(qd, out, dbuf[:, :C], lse, C, B, H, W, heads, L, mode, shift, tables.detach().to(DEV), dbias)
    tol = TOL[dtype] * 4
    close(dqkv, qr.grad, tol, 'dqkv (12288 items)')
    close(dbias, tables.grad, tol * 2, 'dbias tables summed over 4096 windows')


@pytest.mark.parametrize('shape', [(16, 128, 128, 448), (4, 136, 120, 168), (48, 16, 16, 896)])
@pytest.mark.parametrize('dtype', DTYPES)
def test_dwconv_tiled_kernel_timed_shape(dtype, shape):
    """LeFF depthwise 3x3 of decoderlayer_0 in the timed step: B = 16, 128x128, hidden 448 -> 117 M elements >= the 10 M threshold
    of fw_dwconv_fwd / fw_dwconv_bwd, so dwconv_tile_kernel<*, 0> (forward + GELU twin) and dwconv_bwd_fused_kernel (data + weight +
    bias gradient in one pass, persistent over vertically consecutive tiles) run.  Second shape: ragged everywhere -- W = 120 is no
    multiple of the 16-pixel tile, C = 168 leaves a 40-channel last block, 4 x 17 tile rows do not divide by the tiles a workgroup
    walks (workgroups cross image borders).  Third: the smallest tiled layer of the step (one tile per image column)."""
    B, H, W, C = shape
    assert B * H * W * C >= 10_000_000
    h1 = q(rnd(B * H * W, C), dtype).requires_grad_(True)
    w = (rnd(C, 1, 3, 3, seed=1) * 0.3).requires_grad_(True)
    b = (rnd(C, seed=2) * 0.1).requires_grad_(True)
    g1 = q(F.gelu(h1.detach()), dtype)
    ref = F.conv2d(F.gelu(h1).view(B, H, W, C).permute(0, 3, 1, 2), w, b, padding=1, groups=C).permute(0, 2, 3, 1).reshape(B * H * W, C)
    wt = w.detach().view(C, 9).t().contiguous().to(DEV)
    h2, g2 = ops().dwconv_fwd(h1.detach().to(DEV, dtype), wt, b.detach().to(DEV), B, H, W, in_gelu=True)      # the model's path: GELU while the halo tile is staged
    close(h2, ref, TOL[dtype], 'h2 (tiled kernel)')
    close(g2, F.gelu(ref.detach()), TOL[dtype], 'g2 (tiled kernel)')
    dh2 = q(rnd(B * H * W, C, seed=3), dtype)
    ref.backward(dh2)
    dw, db = torch.zeros(C, 9, device=DEV), torch.zeros(C, device=DEV)
    dh1 = ops().dwconv_bwd(dh2.to(DEV, dtype), None, h1.detach().to(DEV, dtype), wt, dw, db, B, H, W)
    close(dh1, h1.grad, TOL[dtype] * 2, 'dh1 (tiled kernel)')
    # 262 144 pixels summed per tap: f32 atomics of block partials; bf16 operands carry 8 bits each
    close(dw, w.grad.view(C, 9), TOL[dtype] * 4, 'dw')
    close(db, b.grad, TOL[dtype] * 4, 'db')


# ------------------------------------------------------------------------------------------------ fused LeFF forward
@pytest.mark.parametrize('C,H,B', [(112, 32, 2), (56, 16, 3), (28, 48, 2)])
def test_leff_fused_forward_matches_unfused_and_reference(C, H, B):
    """csrc/fw_leff.hip (linear1 + GELU + depthwise 3x3 + GELU + linear2 + DropPath + residual in one kernel, bf16) against the
    unfused kernel chain on the same module (outputs, saved twins, every gradient) and against the fp32 PyTorch statement of
    net/utils/leff.py:92-117.  The backward pass is the unfused one in both runs: it consumes the tensors the fused kernel wrote."""
    from fwair import functional as Fn
    from fwair import modules as Mo
    Fn.config.compute_dtype = torch.bfloat16
    Fn.config.direct_grads = False
    torch.manual_seed(C)
    leff = Mo.LeFF(C, 4 * C).to(DEV)
    with torch.no_grad():
        for p_ in leff.parameters():
            p_.copy_(q(p_ * 2.0 + (0.05 if p_.dim() == 1 else 0.0), torch.bfloat16))
    rows = B * H * H
    xn = q(rnd(rows, C), torch.bfloat16)
    res = rnd(rows, C, seed=1)
    scale = torch.tensor([1.0 / 0.9, 0.0, 1.0 / 0.9][:B])
    dy = rnd(rows, C, seed=2)
    outs = []
    keep = Mo._LEFF_FUSED_MIN_ROWS
    try:
        for fused in (True, False):
            Mo._LEFF_FUSED_MIN_ROWS = 0 if fused else 1 << 40
            for p_ in leff.parameters():
                p_.grad = None
            x_ = Fn.act_empty(rows, C, torch.bfloat16, DEV)
            x_.copy_(xn.to(DEV, torch.bfloat16))
            x_.requires_grad_(True)
            r_ = res.to(DEV).requires_grad_(True)
            y = leff.run(x_, r_, scale.to(DEV), B)
            y.backward(dy.to(DEV))
            outs.append((y.detach().cpu(), x_.grad.float().cpu(), r_.grad.cpu(), [p_.grad.detach().cpu().clone() for p_ in leff.parameters()]))
    finally:
        Mo._LEFF_FUSED_MIN_ROWS = keep
        Fn.config.compute_dtype = torch.float32
    (yf, dxf, drf, gf), (yu, dxu, dru, gu) = outs
    close(yf, yu, 4e-3, 'fused y vs unfused y')             # bf16 hidden tensors: GELU inputs rounded at slightly different points
    close(dxf, dxu, 2e-2, 'dx through the tensors the fused kernel saved')
    close(drf, dru, 1e-6, 'd residual')
    for a_, b_, (n_, _) in zip(gf, gu, leff.named_parameters()):
        close(a_, b_, 2e-2, 'grad ' + n_)
    # fp32 statement of the reference
    w1, b1 = leff.linear1[0].weight.detach().cpu(), leff.linear1[0].bias.detach().cpu()
    wd, bd = leff.conv[0].weight.detach().cpu(), leff.conv[0].bias.detach().cpu()
    w2, b2 = leff.linear2[0].weight.detach().cpu(), leff.linear2[0].bias.detach().cpu()
    h = F.gelu(F.linear(xn, w1, b1)).view(B, H, H, 4 * C).permute(0, 3, 1, 2)
    h = F.gelu(F.conv2d(h, wd, bd, padding=1, groups=4 * C)).permute(0, 2, 3, 1).reshape(rows, 4 * C)
    ref = res + F.linear(h, w2, b2) * scale.repeat_interleave(H * H)[:, None]
    close(yf, ref, 2e-2, 'fused y vs fp32 reference')
