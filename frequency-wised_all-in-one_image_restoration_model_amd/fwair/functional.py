"""Autograd shells around the HIP kernels.  PyTorch supplies the tape, the allocator and the stream; every
forward/backward body is a sequence of C-ABI calls (fwair.ops) -- there is no eager-PyTorch arithmetic and no
CPU fallback.  Layout conventions:
  * the residual stream and every parameter / gradient is f32;
  * "T" activations (GEMM operands) are f32 or bf16 according to fwair.config.compute_dtype, stored as
    [tokens, C] views of [tokens, roundup(C, 8)] buffers so that every row starts 16-byte aligned.
"""
import weakref

import torch

from . import ops
from .lib import call, dt


class config:
    compute_dtype = torch.float32       # torch.float32 | torch.bfloat16
    shadow_epoch = 0                    # bumped by the engine after a fused optimizer step
    direct_grads = False                # engine mode: kernels accumulate straight into the (pre-zeroed) flat .grad views


def act_empty(rows, cols, dtype, device):
    """[rows, cols] view of a buffer whose row stride is a multiple of 8 elements."""
    ld = (cols + 7) // 8 * 8
    return torch.empty((rows, ld), dtype=dtype, device=device)[:, :cols]


def aligned(t):
    """Gradients handed over by autograd may be freshly summed contiguous tensors: re-home rows that are not
    16-byte aligned (only possible for bf16 with C % 8 != 0)."""
    if t.stride(1) == 1 and (t.stride(0) * t.element_size()) % 16 == 0 and t.data_ptr() % 16 == 0:
        return t
    out = act_empty(t.shape[0], t.shape[1], t.dtype, t.device)
    out.copy_(t)
    return out


# ---------------------------------------------------------------------------------------------------------------
# low-precision / re-laid-out shadows of parameters
# ---------------------------------------------------------------------------------------------------------------
_shadow = {}          # id(param) -> (weakref(param), {(kind, dtype): (stamp, tensor)}); tensors cannot be dict keys (== is elementwise)


def shadow(param, kind='plain'):
    """Operand view of a parameter in the compute dtype.
    kind: 'plain' [N, K];  'conv4' [Cout, Cin, 4, 4] -> [Cout, 16*Cin] (K order ky, kx, ci);
          'convT2' [Cin, Cout, 2, 2] -> [4*Cout, Cin] (row order i, j, co);  'dw9' [C, 1, 3, 3] -> f32 [9, C]."""
    dtype = config.compute_dtype
    p = param.detach()
    fs = getattr(param, '_fw_shadow', None)
    if fs is not None and kind == 'plain' and fs.dtype == dtype:
        return fs                                   # engine-managed: view of the flat shadow the Adam / EMA kernels write
    if kind == 'plain' and dtype == torch.float32:
        w = p.reshape(p.shape[0], -1)
        if (w.stride(0) * 4) % 16 == 0:
            return w
    slot = _shadow.get(id(param))
    ent = slot[1] if slot is not None and slot[0]() is param else None
    key = (kind, dtype)
    stamp = (param._version, config.shadow_epoch, p.data_ptr())
    if ent is not None and key in ent and ent[key][0] == stamp:
        return ent[key][1]
    if ent is not None and key in ent and ent[key][2] is not None and ent[key][2][0].data_ptr() == p.data_ptr():
        # stale stamp, same parameter storage: re-derive INTO the buffer the entry already owns.  A captured HIP graph (and the
        # fw_permute3_multi table of refresh_shadows) holds this buffer's address; allocating a new one would leave the graph
        # reading freed allocator memory after the next eager forward (an eval between graph replays).
        _, out, recipe = ent[key]
        ops.permute3(recipe[0], out, recipe[1], recipe[2])
        ent[key] = (stamp, out, recipe)
        return out
    recipe = None                                   # (src f32 view, (d0, d1, d2), output strides): out[a*s0 + b*s1 + c*s2] = src[a][b][c]
    if kind == 'plain':
        n, k = p.shape[0], p[0].numel()
        out = act_empty(n, k, dtype, p.device)
        recipe = (p.reshape(n, k), (1, n, k), (0, out.stride(0), 1))
    elif kind == 'conv4':
        co, ci = p.shape[0], p.shape[1]
        out = torch.empty((co, 16 * ci), dtype=dtype, device=p.device)
        recipe = (p.contiguous(), (co, ci, 16), (16 * ci, 1, ci))
    elif kind == 'dw9':
        c = p.shape[0]
        out = torch.empty((9, c), dtype=torch.float32, device=p.device)       # depthwise 3x3 taps, tap-major (always f32)
        recipe = (p.reshape(c, 9), (1, c, 9), (0, 1, c))
    elif kind == 'convT2':
        ci, co = p.shape[0], p.shape[1]
        out = torch.empty((4 * co, ci), dtype=dtype, device=p.device)
        recipe = (p.contiguous(), (ci, co, 4), (1, ci, co * ci))
    elif kind in ('leff1', 'leff2'):
        # operand panels of the fused LeFF kernel (fw_leff_fwd): linear1 [4C][roundup(C, 32)] with zero-padded K;
        # linear2 [roundup(C, 16) + 1][4C] with zero rows (one spare row: the last chunk's fragment reads run 16 elements on)
        n, k = p.shape
        rows, cols = (n, (k + 31) // 32 * 32) if kind == 'leff1' else ((n + 15) // 16 * 16 + 1, k)
        prev = ent[key][1] if ent is not None and key in ent else None
        out = prev if prev is not None and prev.shape == (rows, cols) and prev.device == p.device else \
            torch.zeros((rows, cols), dtype=torch.bfloat16, device=p.device)     # re-cast in place: captured graphs hold the address
        call('fw_cast_rows', 1, p, k, out, cols, n, k, None, 1)
    else:
        raise ValueError(kind)
    if recipe is not None:
        src, dims, strides = recipe
        assert src.data_ptr() == p.data_ptr() and src.is_contiguous(), 'shadow sources must alias the parameter (refresh_shadows re-reads them)'
        ops.permute3(src, out, dims, strides)
    if ent is None:
        ent = {}
        pid = id(param)
        _shadow[pid] = (weakref.ref(param, lambda _r, pid=pid: _shadow.pop(pid, None)), ent)
    ent[key] = (stamp, out, recipe)
    return out


def _shadow_part(param, key, out, recipe):
    """Keep `out` (a view into a buffer shared with other parameters) as the cached copy `key` of `param`: filled now, re-derived by
    refresh_shadows / on a stale stamp like every other entry."""
    p = param.detach()
    slot = _shadow.get(id(param))
    ent = slot[1] if slot is not None and slot[0]() is param else None
    stamp = (param._version, config.shadow_epoch, p.data_ptr())
    if ent is not None and key in ent and ent[key][0] == stamp and ent[key][1].data_ptr() == out.data_ptr():
        return
    src, dims, strides = recipe
    assert src.data_ptr() == p.data_ptr() and src.is_contiguous()
    ops.permute3(src, out, dims, strides)
    if ent is None:
        ent = {}
        pid = id(param)
        _shadow[pid] = (weakref.ref(param, lambda _r, pid=pid: _shadow.pop(pid, None)), ent)
    ent[key] = (stamp, out, recipe)


_qkv_cat = {}                 # id(wq) -> (weakref(wq), {dtype: (W [Cp + 2C][ld], bias f32 [Cp + 2C])})


def shadow_qkv(wq, bq, wkv, bkv):
    """to_q / to_kv whose width C is not a multiple of 8 (the C = 28 stage): ONE operand [Cp + 2C][K] = [Wq ; 0 ; Wkv] and one bias
    [bq ; 0 ; bkv], so that the projection writes whole rows of the q | pad | k | v buffer in one GEMM (two GEMMs wrote 56- and
    112-byte pieces of every 176-byte row: 65 + 72 us against 45 us for the same bytes at 786 432 tokens) and its input gradient
    reads the whole gradient buffer in one.  The pad rows are zeroed once; the two halves are ordinary shadow entries of their
    parameters (functional.refresh_shadows re-derives them after every optimizer step)."""
    dtype = config.compute_dtype
    C, K = wq.shape
    Cp = (C + 7) // 8 * 8
    slot = _qkv_cat.get(id(wq))
    bufs = slot[1] if slot is not None and slot[0]() is wq else None
    if bufs is None:
        bufs = {}
        pid = id(wq)
        _qkv_cat[pid] = (weakref.ref(wq, lambda _r, pid=pid: _qkv_cat.pop(pid, None)), bufs)
    if dtype not in bufs:
        ld = (K + 7) // 8 * 8
        bufs[dtype] = (torch.zeros((Cp + 2 * C, ld), dtype=dtype, device=wq.device)[:, :K],
                       torch.zeros((Cp + 2 * C,), dtype=torch.float32, device=wq.device))
    W, b = bufs[dtype]
    ld = W.stride(0)
    _shadow_part(wq, ('qcat', dtype), W[:C], (wq.detach().reshape(C, K), (1, C, K), (0, ld, 1)))
    _shadow_part(wkv, ('kvcat', dtype), W[Cp:], (wkv.detach().reshape(2 * C, K), (1, 2 * C, K), (0, ld, 1)))
    _shadow_part(bq, ('qcat', torch.float32), b[:C], (bq.detach().reshape(1, C), (1, 1, C), (0, 0, 1)))
    _shadow_part(bkv, ('kvcat', torch.float32), b[Cp:], (bkv.detach().reshape(1, 2 * C), (1, 1, 2 * C), (0, 0, 1)))
    return W, b


_refresh_tables = {}          # which -> (signature, device table, device prefix, num, total blocks)
_retired_tables = []          # superseded tables, kept alive: captured graphs hold their addresses


def refresh_shadows(which='all', only_ids=None, restamp_others=False):
    """Engine hook: the parameters were just rewritten in place (fused Adam / EMA kernels) and config.shadow_epoch bumped.  Re-derive
    EVERY cached re-laid-out operand copy (depthwise taps, convolution weights, unaligned bf16 rows: ~170 per model) with ONE
    fw_permute3_multi launch into the buffers they already live in, and mark them fresh -- instead of one 5-8 us launch each when the
    next forward asks for them.  only_ids: restrict to the parameters with these ids (the key encoder after its EMA update);
    restamp_others: the remaining entries are known to be unchanged and are marked fresh without a launch.
    The table of a set is built on first use OUTSIDE stream capture; inside a capture an unseen set is left to the per-entry path."""
    items, stamps = [], []
    for pid, (ref, ent) in list(_shadow.items()):                    # a weakref callback may pop entries while we walk (cyclic GC)
        param = ref()
        if param is None:
            continue
        mine = only_ids is None or pid in only_ids
        for key, (stamp, out, recipe) in list(ent.items()):
            if recipe is None or not (mine or restamp_others):
                continue
            # entries that are only re-stamped keep their recorded tensor version: an in-place update that did not go through
            # the engine (load_state_dict, a torch optimizer) still shows as stale and is re-derived by shadow()
            fresh = (param._version if mine else stamp[0], config.shadow_epoch, param.data_ptr())
            if fresh[2] != recipe[0].data_ptr():
                continue                                             # the parameter moved (e.g. .to(device)): let shadow() rebuild it
            stamps.append((ent, key, fresh, out, recipe))
            if mine:
                src, dims, strides = recipe
                items.append((src.data_ptr(), out.data_ptr(), dims[0], dims[1], dims[2], strides[0], strides[1], strides[2],
                              int(out.dtype == torch.bfloat16), 0))
    if items:
        sig = tuple(items)
        cached = _refresh_tables.get(which)
        if cached is None or cached[0] != sig:
            if torch.cuda.is_current_stream_capturing():
                # no host -> device table upload inside a capture, and a graph without the refresh would replay on operand
                # copies frozen at capture time: fail the capture loudly (the engine warms up eagerly first, which builds it)
                raise RuntimeError(f'fwair: refresh_shadows({which!r}) has no device table for this parameter set inside a stream '
                                   'capture; run one eager step before capturing')
            offs, total = [0], 0
            for it in items:
                total += (it[2] * it[3] * it[4] + 1023) // 1024
                offs.append(total)
            dev = stamps[0][3].device
            cached = (sig, torch.tensor(items, dtype=torch.int64).to(dev), torch.tensor(offs, dtype=torch.int64).to(dev), len(items), total)
            if _refresh_tables.get(which) is not None:
                _retired_tables.append(_refresh_tables[which])       # a captured graph may still launch from the superseded table
            _refresh_tables[which] = cached
        call('fw_permute3_multi', cached[1], cached[2], cached[3], cached[4])
    for ent, key, fresh, out, recipe in stamps:
        ent[key] = (fresh, out, recipe)


def _zeros(shape, device):
    return torch.zeros(shape, dtype=torch.float32, device=device)


def _grad_target(param, shape=None):
    """(buffer to accumulate into, value to hand back to autograd).  In engine mode the kernels add straight into the
    parameter's flat .grad view and autograd gets None (no zero-fill, no extra add kernel)."""
    if config.direct_grads and param is not None and param.grad is not None:
        g = param.grad
        return (g if shape is None else g.view(shape)), None
    z = _zeros(param.shape if shape is None else shape, param.device)
    return z, z


def _wgrad(g, x, n, k, m, weight, bias=None):
    """dW[n][k] += sum_m g[m][n] x[m][k]  (both operands reduction-major); db[n] += sum_m g[m][n] falls out of the
    same pass (xsum).  Returns the autograd values (dW, db)."""
    dw, rw = _grad_target(weight, (n, k))
    db, rb = _grad_target(bias) if bias is not None else (None, None)
    direct = rw is None and rb is None                 # persistent .grad views (engine mode: zeroed before every pass): autograd copies what it is handed otherwise
    ops.wgrad(g, x, n, k, m, dw, db, defer=direct, fresh=direct)
    return (rw.view_as(weight) if rw is not None else None), rb


def _bgrad(g, bias):
    db, rb = _grad_target(bias)
    ops.colsum(g, db)
    return rb


# ---------------------------------------------------------------------------------------------------------------
# LayerNorm : f32 stream -> T
# ---------------------------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        rows, C = x.shape
        y = act_empty(rows, C, config.compute_dtype, x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        call('fw_layernorm_fwd', dt(y.dtype), x, x.stride(0), gamma, beta, y, y.stride(0), mean, rstd, rows, C, 1e-5)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.beta = beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = aligned(dy)
        beta = ctx.beta
        dg, rg = _grad_target(gamma)
        db, rb = _grad_target(beta)
        dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, defer=rg is None and rb is None)
        return dx, rg, rb


# ---------------------------------------------------------------------------------------------------------------
# Linear : T -> T | f32 (+ residual stream, DropPath row scale, GELU on the input, LeakyReLU on the output)
# ---------------------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = [residual +] rowscale * (x W^T + b).   x: T [M, K];  W: f32 parameter [N, K].
    gelu_out: also return g = GELU(y) (non-differentiable twin; LeFF keeps pre- and post-activations).
    x_pre: x is GELU(x_pre) computed upstream; the input gradient is routed to x_pre (times GELU'(x_pre)).
    out_f32: y is f32."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, rowscale, rows_per_scale, x_pre, gelu_out, out_f32, fuse=None):
        ctx.set_materialize_grads(False)            # the GELU twin never gets a gradient: do not zero-fill one for it
        M, K = x.shape
        N = weight.shape[0]
        if fuse is not None and not gelu_out:       # linear2 of a fused LeFF: fw_leff_fwd already produced y (see `leff_fused`)
            y, g = fuse['y'], None
        elif fuse is not None:                      # linear1 of a fused LeFF: ONE kernel computes the whole feed-forward
            y, g = _leff_fused_forward(x, weight, bias, fuse)
        else:
            w = shadow(weight)
            if residual is not None or out_f32:
                y = torch.empty((M, N), dtype=torch.float32, device=x.device)
            else:
                y = act_empty(M, N, x.dtype, x.device)
            g = act_empty(M, N, x.dtype, x.device) if gelu_out else None
            ops.gemm(x, w, M, N, K, out=y, bias=bias, rowscale=rowscale, rows_per_scale=rows_per_scale, residual=residual,
                     out_gelu=g)
        ctx.save_for_backward(x, weight, rowscale, x_pre)
        ctx.bias = bias
        ctx.cfg = (rows_per_scale, residual is not None)
        if gelu_out:
            ctx.mark_non_differentiable(g)
            return y, g
        return y

    @staticmethod
    def backward(ctx, dy, *_):
        if dy is None:
            return (None,) * 10
        x, weight, rowscale, x_pre = ctx.saved_tensors
        rows_per_scale, has_res = ctx.cfg
        bias = ctx.bias
        M, K = x.shape
        N = weight.shape[0]
        tw = getattr(dy, '_fw_twin', None)
        if (tw is not None and tw[0].dtype == x.dtype and tw[0].shape == (M, N) and tw[2] == rows_per_scale
                and (tw[1] is rowscale or (tw[1] is not None and rowscale is not None and tw[1].data_ptr() == rowscale.data_ptr()))):
            g = tw[0]                                        # LnResFn.backward already wrote T(dy * rowscale)
        elif dy.dtype == torch.float32 and x.dtype != torch.float32 or rowscale is not None:
            g = act_empty(M, N, x.dtype, x.device)
            call('fw_cast_rows', dt(x.dtype), dy, dy.stride(0), g, g.stride(0), M, N, rowscale, rows_per_scale)
        else:
            g = aligned(dy)
        dw, db = _wgrad(g, x, N, K, M, weight, bias)
        dx = dpre = None
        if ctx.needs_input_grad[0] or (x_pre is not None and ctx.needs_input_grad[6]):
            d = act_empty(M, K, x.dtype, x.device)
            ops.dgrad(g, shadow(weight), M, K, N, d, act=2 if x_pre is not None else 0, aux=x_pre)
            if x_pre is not None:
                dpre = d
            else:
                dx = d
        return dx, dw, db, (dy if has_res else None), None, None, dpre, None, None, None


def _leff_fused_forward(xn, w1, b1, fuse):
    """fw_leff_fwd: y = res + rowscale * linear2(GELU(dwconv(GELU(linear1(xn))))) in one kernel; returns (h1, g1) -- the outputs
    of the linear1 node -- and leaves h2, g2, y in `fuse` for the DwConvFn / LinearFn nodes that follow (they keep their own
    backward; only their forward launches are replaced)."""
    M, C = xn.shape
    B, H, W = fuse['geo']
    C4 = w1.shape[0]
    h1, g1 = act_empty(M, C4, xn.dtype, xn.device), act_empty(M, C4, xn.dtype, xn.device)
    h2, g2 = act_empty(M, C4, xn.dtype, xn.device), act_empty(M, C4, xn.dtype, xn.device)
    y = torch.empty((M, C), dtype=torch.float32, device=xn.device)
    res, rs = fuse['residual'], fuse['rowscale']
    call('fw_leff_fwd', xn, xn.stride(0), shadow(w1, 'leff1'), b1, shadow(fuse['wd'], 'dw9'), fuse['bd'], shadow(fuse['w2'], 'leff2'), fuse['b2'],
         res, res.stride(0), rs, fuse['rows_per_scale'], y, y.stride(0), h1, g1, h2, g2, h1.stride(0), B, H, W, C)
    fuse['h2'], fuse['g2'], fuse['y'] = h2, g2, y
    return h1, g1


class LnResFn(torch.autograd.Function):
    """(x) -> (x, LayerNorm(x)): the stream feeds both the branch and the residual add; returning it as a second output
    lets backward fuse  dx = d_residual + LN'(d_branch)  into the LayerNorm kernel instead of a separate add."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        ctx.set_materialize_grads(False)
        ctx.rs_hint = getattr(x, '_fw_rs', None)            # x came straight out of a residual Linear: (rowscale, rows_per_scale)
        rows, C = x.shape
        y = act_empty(rows, C, config.compute_dtype, x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        call('fw_layernorm_fwd', dt(y.dtype), x, x.stride(0), gamma, beta, y, y.stride(0), mean, rstd, rows, C, 1e-5)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.beta = beta
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dg, rg = _grad_target(gamma)
        db, rb = _grad_target(ctx.beta)
        if dy is None:
            return dres, None, None
        dy = aligned(dy)
        twin = None
        if ctx.rs_hint is not None:
            # the gradient returned here is the dy of the Linear that produced x; its backward wants T(dy * DropPath scale) as the
            # operand of two GEMMs -- written by this kernel as a second output instead of a separate cast pass
            twin = act_empty(x.shape[0], x.shape[1], dy.dtype, x.device)
        rs, rps = ctx.rs_hint if ctx.rs_hint is not None else (None, 1)
        dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dres=dres, defer=rg is None and rb is None, twin=twin, twscale=rs,
                               tw_rows_per_scale=rps)
        if twin is not None:
            dx._fw_twin = (twin, rs, rps)
        return dx, rg, rb


def linear(x, weight, bias=None, residual=None, rowscale=None, rows_per_scale=1, x_pre=None, gelu_out=False, out_f32=False, fuse=None):
    y = LinearFn.apply(x, weight, bias, residual, rowscale, rows_per_scale, x_pre, gelu_out, out_f32, fuse)
    if residual is not None and y.requires_grad and (x.dtype != torch.float32 or rowscale is not None):
        y._fw_rs = (rowscale, rows_per_scale)               # lets the LayerNorm that consumes y emit this Linear's backward operand
    return y


class QKVFn(torch.autograd.Function):
    """qkv buffer [M, Cp + 2C] = [ x Wq^T + bq | pad | x Wkv^T + bkv ],  Cp = roundup(C, 8)
    (decoder_Uformer.py:121-124: to_q and to_kv outputs, k = kv[:, :C], v = kv[:, C:]).
    fused: the engine's single [3C, C] view of both weights (engine.TrainEngine._fuse_projections) -> one GEMM each way."""

    @staticmethod
    def forward(ctx, x, wq, bq, wkv, bkv, fused):
        M, K = x.shape
        C = wq.shape[0]
        Cp = (C + 7) // 8 * 8
        buf = act_empty(M, Cp + 2 * C, x.dtype, x.device)
        if fused is not None and Cp == C:
            w3 = fused['sw'] if x.dtype == torch.bfloat16 and 'sw' in fused else (fused['w'] if x.dtype == torch.float32 else None)
        else:
            w3 = None
        ctx.cat = None
        if w3 is not None:
            ops.gemm(x, w3, M, 3 * C, K, out=buf[:, :3 * C], bias=fused['b'])
        elif Cp != C and bq is not None and bkv is not None:
            wc, bc = shadow_qkv(wq, bq, wkv, bkv)             # [Wq ; 0 ; Wkv]: whole rows of the buffer (the pad columns become 0)
            ops.gemm(x, wc, M, Cp + 2 * C, K, out=buf, bias=bc)
            ctx.cat = wc
        else:
            ops.gemm(x, shadow(wq), M, C, K, out=buf[:, :C], bias=bq)
            ops.gemm(x, shadow(wkv), M, 2 * C, K, out=buf[:, Cp:], bias=bkv)
        ctx.save_for_backward(x, wq, wkv)
        ctx.bq, ctx.bkv = bq, bkv
        ctx.fused = (fused, w3)
        return buf

    @staticmethod
    def backward(ctx, dbuf):
        x, wq, wkv = ctx.saved_tensors
        M, K = x.shape
        C = wq.shape[0]
        Cp = (C + 7) // 8 * 8
        dbuf = aligned(dbuf)
        fused, w3 = ctx.fused
        dx = act_empty(M, K, x.dtype, x.device)
        if w3 is not None and config.direct_grads and 'gw' in fused:
            ops.wgrad(dbuf[:, :3 * C], x, 3 * C, K, M, fused['gw'], fused['gb'], defer=True, fresh=True)
            ops.dgrad(dbuf[:, :3 * C], w3, M, K, 3 * C, dx)          # split over the 3C reduction when the output has few tiles
            return dx, None, None, None, None, None
        dq, dkv = dbuf[:, :C], dbuf[:, Cp:]
        dwq, dbq = _wgrad(dq, x, C, K, M, wq, ctx.bq)
        dwkv, dbkv = _wgrad(dkv, x, 2 * C, K, M, wkv, ctx.bkv)
        if ctx.cat is not None and getattr(dbuf, '_fw_zero_pad', False):
            # the attention backward wrote zeros into the pad columns (fw_attn_bwd dq_pad): the whole buffer is one operand
            ops.gemm(dbuf, ctx.cat, M, K, Cp + 2 * C, w_trans=True, out=dx)
            return dx, dwq, dbq, dwkv, dbkv, None
        tmp = torch.empty((M, K), dtype=torch.float32, device=x.device)
        ops.gemm(dq, shadow(wq), M, K, C, w_trans=True, out=tmp)
        ops.gemm(dkv, shadow(wkv), M, K, 2 * C, w_trans=True, out=dx, residual=tmp)
        return dx, dwq, dbq, dwkv, dbkv, None


# ---------------------------------------------------------------------------------------------------------------
# window attention
# ---------------------------------------------------------------------------------------------------------------
class WindowAttnFn(torch.autograd.Function):
    """qkv: buffer of QKVFn.  tables: f32 [ntab, 225, heads].  coef: f32 [B, heads, 3] or None.
    tparam: the Parameter `tables` is a view of (engine mode: its flat .grad view takes the kernel's atomics directly).
    dcoef_to: f32 [B, heads, 3] slice of the coefficient producer's gradient buffer -- the kernel accumulates there and
    autograd gets None (a slice gradient would cost a full-size zero-fill + add per block)."""

    @staticmethod
    def forward(ctx, qkv, tables, coef, geo, tparam, dcoef_to):
        ctx.side = (tparam, dcoef_to)
        C, B, H, W, heads, L, mode, shift, lfs = geo
        D = C // heads
        Cp = (C + 7) // 8 * 8
        rows = qkv.shape[0]
        nkt = 1 if mode == 0 else L - 1
        out = act_empty(rows, C, qkv.dtype, qkv.device)
        lse = torch.empty((B * (H // 8) * (W // 8) * L * heads, 64), dtype=torch.float32, device=qkv.device)
        tab = ops._lfs.device_table(qkv.dtype, qkv.device) if lfs == 2 else None
        call('fw_attn_fwd', dt(qkv.dtype), D, nkt, lfs, qkv, qkv[:, Cp:], qkv[:, Cp + C:], qkv.stride(0), out, out.stride(0),
             lse, tables, coef, tab, B, H, W, heads, L, mode, shift, float(D) ** -0.5)
        ctx.save_for_backward(qkv, tables, coef, out, lse)
        ctx.geo = geo
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, tables, coef, out, lse = ctx.saved_tensors
        C, B, H, W, heads, L, mode, shift, lfs = ctx.geo
        D = C // heads
        Cp = (C + 7) // 8 * 8
        rows = qkv.shape[0]
        nkt = 1 if mode == 0 else L - 1
        dout = aligned(dout)
        dqkv = act_empty(rows, qkv.shape[1], qkv.dtype, qkv.device)
        d2 = act_empty(rows, qkv.shape[1], qkv.dtype, qkv.device) if nkt == 2 else None
        tparam, dcoef_to = ctx.side
        if config.direct_grads and tparam is not None and tparam.grad is not None:
            dtab, rtab = tparam.grad.view(tables.shape), None
        else:
            dtab = rtab = _zeros(tables.shape, qkv.device)
        if coef is None:
            dcoef = rcoef = None
        elif dcoef_to is not None:
            dcoef, rcoef = dcoef_to, None
        else:
            dcoef = rcoef = _zeros(coef.shape, qkv.device)
        tab = ops._lfs.device_table(qkv.dtype, qkv.device) if lfs == 2 else None
        call('fw_attn_bwd', dt(qkv.dtype), D, nkt, lfs, qkv, qkv[:, Cp:], qkv[:, Cp + C:], qkv.stride(0), out, out.stride(0),
             dout, dout.stride(0), lse, tables, coef, tab, dqkv, dqkv[:, Cp:], dqkv[:, Cp + C:],
             d2[:, Cp:] if d2 is not None else None, d2[:, Cp + C:] if d2 is not None else None, dqkv.stride(0), dtab, dcoef,
             B, H, W, heads, L, mode, shift, float(D) ** -0.5, Cp - C)      # the pad between dq and dk gets zeros: dqkv is ONE GEMM operand
        if nkt == 2:
            call('fw_add_rows', dt(qkv.dtype), d2[:, Cp:], d2.stride(0), dqkv[:, Cp:], dqkv.stride(0), rows, 2 * C)
        dqkv._fw_zero_pad = True                               # q | 0 | k | v: defined everywhere
        return dqkv, rtab, rcoef, None, None, None


# ---------------------------------------------------------------------------------------------------------------
# LeFF depthwise conv (input and output are pre-activations; GELU is applied on load by the consumers)
# ---------------------------------------------------------------------------------------------------------------
class DwConvFn(torch.autograd.Function):
    """(h1[, g1 = GELU(h1)]) -> (h2, g2 = GELU(h2)).  Gradients flow through the pre-activations only: backward receives
    d h2 (LinearFn routes it there via x_pre) and returns d h1.  g1 = None (the default path): GELU(h1) is evaluated inside the
    kernels as the input tile is staged -- forward AND weight gradient -- so that activation never travels through HBM."""

    @staticmethod
    def forward(ctx, h1, g1, weight, bias, B, H, W, fuse=None):
        ctx.set_materialize_grads(False)
        C = h1.shape[1]
        wt = shadow(weight, 'dw9')                                               # tap-major copy of the [C,1,3,3] weight, cached per step
        if fuse is not None:                                                     # fused LeFF forward: fw_leff_fwd already wrote both
            h2, g2 = fuse['h2'], fuse['g2']
        else:
            h2 = act_empty(h1.shape[0], C, h1.dtype, h1.device)
            g2 = act_empty(h1.shape[0], C, h1.dtype, h1.device)
            src = g1 if g1 is not None else h1
            call('fw_dwconv_fwd', dt(h1.dtype), src, src.stride(0), int(g1 is None), wt, bias, h2, g2, h2.stride(0), B, H, W, C)
        if g1 is not None:
            ctx.save_for_backward(h1, weight, wt, g1)
        else:
            ctx.save_for_backward(h1, weight, wt)
        ctx.bias = bias
        ctx.geo = (B, H, W)
        ctx.mark_non_differentiable(g2)
        return h2, g2

    @staticmethod
    def backward(ctx, dh2, _):
        if dh2 is None:
            return (None,) * 8
        h1, weight, wt = ctx.saved_tensors[:3]
        g1 = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        B, H, W = ctx.geo
        C = h1.shape[1]
        dh2 = aligned(dh2)
        db, rb = _grad_target(ctx.bias)
        dw, rw = _grad_target(weight, (C, 9))                                    # the kernel adds in the parameter's layout
        dh1 = act_empty(h1.shape[0], C, h1.dtype, h1.device)
        call('fw_dwconv_bwd', dt(h1.dtype), dh2, dh2.stride(0), g1, h1, h1.stride(0), wt, dh1, dh1.stride(0), dw, db, B, H, W, C)
        return dh1, None, (rw.view_as(weight) if rw is not None else None), rb, None, None, None, None


# ---------------------------------------------------------------------------------------------------------------
# convolutions
# ---------------------------------------------------------------------------------------------------------------
class DownsampleFn(torch.autograd.Function):
    """Conv2d k4 s2 p1 on the token stream (decoder_Uformer.py:423-430): im2col + GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, B, H, W):
        C, Co = x.shape[1], weight.shape[0]
        col = ops.im2col4(x, B, H, W, config.compute_dtype)
        y = torch.empty((col.shape[0], Co), dtype=torch.float32, device=x.device)
        ops.gemm(col, shadow(weight, 'conv4'), col.shape[0], Co, 16 * C, out=y, bias=bias)
        ctx.save_for_backward(x, weight)
        ctx.bias = bias
        ctx.geo = (B, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, H, W = ctx.geo
        C, Co = x.shape[1], weight.shape[0]
        Mo = dy.shape[0]
        g = ops.cast_rows(dy, config.compute_dtype) if config.compute_dtype != torch.float32 else aligned(dy)
        col = ops.im2col4(x, B, H, W, config.compute_dtype)
        dwk = _zeros((Co, 16 * C), x.device)
        db, rb = _grad_target(ctx.bias)
        ops.wgrad(g, col, Co, 16 * C, Mo, dwk, db)
        dw, rw = _grad_target(weight, (Co, C, 16))
        ops.permute3(dwk, dw, (Co, 16, C), (16 * C, 1, 16), accumulate=True)
        del col
        dcol = ops.gemm(g, shadow(weight, 'conv4'), Mo, 16 * C, Co, w_trans=True)
        dx = ops.col2im4(dcol, B, H, W, C)
        return dx, (rw.view_as(weight) if rw is not None else None), rb, None, None, None


class UpsampleCatFn(torch.autograd.Function):
    """ConvTranspose2d k2 s2 (decoder_Uformer.py:443-449) followed by torch.cat([up, skip], -1) (:1162)."""

    @staticmethod
    def forward(ctx, x, skip, weight, bias, B, H, W):
        Cin, Cout = weight.shape[0], weight.shape[1]
        Cs = skip.shape[1]
        xq = ops.cast_rows(x, config.compute_dtype) if config.compute_dtype != torch.float32 else x
        g = ops.gemm(xq, shadow(weight, 'convT2'), x.shape[0], 4 * Cout, Cin)
        out = torch.empty((4 * x.shape[0], Cout + Cs), dtype=torch.float32, device=x.device)
        ops.pixel_shuffle(g, bias, out[:, :Cout], B, H, W, Cout)
        ops.copy_rows(skip, out[:, Cout:])
        ctx.save_for_backward(x, weight)
        ctx.bias = bias
        ctx.geo = (B, H, W, Cs)
        return out

    @staticmethod
    def backward(ctx, dcat):
        x, weight = ctx.saved_tensors
        B, H, W, Cs = ctx.geo
        Cin, Cout = weight.shape[0], weight.shape[1]
        M = x.shape[0]
        dg = ops.pixel_unshuffle(dcat[:, :Cout], B, H, W, Cout, config.compute_dtype)
        xq = ops.cast_rows(x, config.compute_dtype) if config.compute_dtype != torch.float32 else x
        dwt = _zeros((4 * Cout, Cin), x.device)
        ops.wgrad(dg, xq, 4 * Cout, Cin, M, dwt)
        dw, rw = _grad_target(weight, (Cin, Cout, 4))
        ops.permute3(dwt, dw, (4, Cout, Cin), (1, 4, Cout * 4), accumulate=True)
        dx = ops.gemm(dg, shadow(weight, 'convT2'), M, Cin, 4 * Cout, w_trans=True, out_dtype=torch.float32)
        db, rb = _grad_target(ctx.bias)
        ops.colsum(dcat[:, :Cout], db)
        dskip = torch.empty((dcat.shape[0], Cs), dtype=torch.float32, device=x.device)
        ops.copy_rows(dcat[:, Cout:], dskip)
        return dx, dskip, (rw.view_as(weight) if rw is not None else None), rb, None, None, None


class InputProjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, weight, bias):
        B, _, H, W = img.shape
        C = weight.shape[0]
        img = img.contiguous()
        out = torch.empty((B * H * W, C), dtype=torch.float32, device=img.device)
        call('fw_inproj_fwd', img, weight, bias, out, C, B, H, W, C, 0.01)
        ctx.save_for_backward(img, weight, out)
        return out

    @staticmethod
    def backward(ctx, dy):
        img, weight, out = ctx.saved_tensors
        B, _, H, W = img.shape
        C = weight.shape[0]
        dw, db = _zeros(weight.shape, img.device), _zeros((C,), img.device)
        dy = dy.contiguous()
        call('fw_inproj_bwd', img, out, C, dy, dy.stride(0), dw, db, B, H, W, C, 0.01)
        return None, dw, db


class OutputProjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fea, weight, bias, img):
        B, _, H, W = img.shape
        C = fea.shape[1]
        img = img.contiguous()
        out = torch.empty_like(img)
        call('fw_outproj_fwd', fea, fea.stride(0), weight, bias, img, out, B, H, W, C)
        ctx.save_for_backward(fea, weight)
        ctx.geo = (B, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        fea, weight = ctx.saved_tensors
        B, H, W = ctx.geo
        C = fea.shape[1]
        dout = dout.contiguous()
        dfea = torch.empty((fea.shape[0], C), dtype=torch.float32, device=fea.device)
        dw, db = _zeros(weight.shape, fea.device), _zeros((3,), fea.device)
        call('fw_outproj_bwd', dout, fea, fea.stride(0), weight, dfea, C, dw, db, B, H, W, C)
        return dfea, dw, db, None


# ---------------------------------------------------------------------------------------------------------------
# encoder contrastive head
# ---------------------------------------------------------------------------------------------------------------
class BnLreluGapFn(torch.autograd.Function):
    """fea: T [B*(S/16)^2, ED*256] viewed as [B][ED][P] -> BatchNorm2d -> LeakyReLU(0.1) -> mean over P  (encoder_Uformer.py:978-982)."""

    @staticmethod
    def forward(ctx, fea, gamma, beta, rmean, rvar, nbt, B, training):
        ED = gamma.shape[0]
        P = fea.numel() // (B * ED)
        assert fea.is_contiguous()
        dev = fea.device
        gap = torch.empty((B, ED), dtype=torch.float32, device=dev)
        part = torch.empty((ED, B, 2), dtype=torch.float32, device=dev) if training else None
        saved = torch.empty((ED, 2), dtype=torch.float32, device=dev) if training else None
        call('fw_bn_lrelu_gap_fwd', dt(fea.dtype), fea, gamma, beta, rmean, rvar, nbt, part, saved, gap, B, ED, P, int(training),
             1e-5, 0.1, 0.1)
        ctx.save_for_backward(fea, gamma, beta, saved)
        ctx.geo = (B, ED, P, training)
        return gap

    @staticmethod
    def backward(ctx, dgap):
        fea, gamma, beta, saved = ctx.saved_tensors
        B, ED, P, training = ctx.geo
        assert training, 'backward through eval-mode BatchNorm is not part of the hot path'
        dev = fea.device
        part2 = torch.empty((ED, B, 2), dtype=torch.float32, device=dev)
        dfea = torch.empty_like(fea)
        dg, db = _zeros((ED,), dev), _zeros((ED,), dev)
        call('fw_bn_lrelu_gap_bwd', dt(fea.dtype), fea, gamma, beta, saved, dgap.contiguous(), part2, dfea, dg, db, B, ED, P, 0.1)
        return dfea, dg, db, None, None, None, None, None


class LreluFn(torch.autograd.Function):
    """f32 -> T LeakyReLU (head MLPs)."""

    @staticmethod
    def forward(ctx, x, slope):
        x = x.contiguous()
        y = act_empty(x.shape[0], x.shape[1], config.compute_dtype, x.device)
        if y.stride(0) != x.shape[1]:
            y = torch.empty(x.shape, dtype=config.compute_dtype, device=x.device)
        call('fw_lrelu_fwd', dt(y.dtype), x, y, x.numel(), slope)
        ctx.save_for_backward(x)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        call('fw_lrelu_bwd', dt(dy.dtype), dy, x, dx, x.numel(), ctx.slope)
        return dx, None


class CastFn(torch.autograd.Function):
    """f32 -> T (identity in f32 mode); backward T -> f32."""

    @staticmethod
    def forward(ctx, x):
        if config.compute_dtype == torch.float32:
            return x.view_as(x)
        return ops.cast_rows(x, config.compute_dtype)

    @staticmethod
    def backward(ctx, dy):
        if dy.dtype == torch.float32:
            return dy
        dy = dy.contiguous()
        out = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
        ops.permute3(dy, out, (1, 1, dy.numel()), (0, 0, 1))
        return out


# ---------------------------------------------------------------------------------------------------------------
# MoCo logits
# ---------------------------------------------------------------------------------------------------------------
class MocoLogitsFn(torch.autograd.Function):
    """q, k: f32 [L, B, ED] (un-normalised); queue: f32 [L, ED, K] -> logits [L, B, 1+K] (moco.py:127-156)."""

    @staticmethod
    def forward(ctx, q, k, queue, T):
        L, B, ED = q.shape
        K = queue.shape[2]
        q, k = q.contiguous(), k.contiguous()
        qsnap = queue.clone()                      # moco.py:149 clones the queue before it is overwritten
        logits = torch.empty((L, B, 1 + K), dtype=torch.float32, device=q.device)
        khat = torch.empty((L, B, ED), dtype=torch.float32, device=q.device)
        call('fw_moco_logits', q, k, qsnap, logits, khat, L, B, ED, K, 1.0 / T)
        ctx.save_for_backward(q, khat, qsnap)
        ctx.T = T
        ctx.mark_non_differentiable(khat)
        return logits, khat

    @staticmethod
    def backward(ctx, dlogits, _):
        q, khat, qsnap = ctx.saved_tensors
        L, B, ED = q.shape
        dq = torch.empty_like(q)
        call('fw_moco_logits_bwd', q, khat, qsnap, dlogits.contiguous(), dq, L, B, ED, qsnap.shape[2], 1.0 / ctx.T)
        return dq, None, None, None


# ---------------------------------------------------------------------------------------------------------------
# LFS lambda heads of all decoder blocks in one launch
# ---------------------------------------------------------------------------------------------------------------
_lfs_tables = {}


class LfsLambdaFn(torch.autograd.Function):
    """inter: f32 [nb1*B*NT, C] (bands 1.. of the encoder output).  params: per block, per band, the 8 tensors
    (ln.w, ln.b, lin.w, lin.b, mlp0.w, mlp0.b, mlp2.w, mlp2.b).  Returns the flat (a, b, c) coefficient buffer."""

    @staticmethod
    def forward(ctx, inter, meta, *params):
        ctx.set_materialize_grads(False)
        heads_list, B, nb1 = meta
        dev = inter.device
        C = inter.shape[1]
        nblk = len(heads_list)
        inter = inter.contiguous()
        xbar = torch.empty((nb1 * B, C), dtype=torch.float32, device=dev)
        NT = inter.shape[0] // (nb1 * B)                  # tokens of the encoder representation: (S/16)^2, 64 at 128x128
        ctx.NT = NT
        stats = torch.empty((nb1 * B, NT, 2), dtype=torch.float32, device=dev)
        call('fw_lfs_xbar', inter, xbar, stats, nb1, B, NT, C, 1e-5)
        key = (tuple(p.data_ptr() for p in params), B, nb1, str(dev))
        tabs = _lfs_tables.get(key)
        if tabs is None:                    # host-built once per parameter placement (never inside a graph capture)
            ptab = torch.zeros((nblk, 2, 8), dtype=torch.int64)
            it = iter(params)
            for bi in range(nblk):
                for band in range(nb1):
                    for j in range(8):
                        ptab[bi, band, j] = next(it).data_ptr()
            offs = [0]
            for h in heads_list:
                offs.append(offs[-1] + B * h * 3)
            total = sum(p.numel() for p in params)
            gflat = torch.zeros(total, dtype=torch.float32, device=dev)
            gt, o, k = torch.zeros((nblk, 2, 8), dtype=torch.int64), 0, 0
            for bi in range(nblk):
                for band in range(nb1):
                    for j in range(8):
                        gt[bi, band, j] = gflat.data_ptr() + 4 * o
                        o += params[k].numel()
                        k += 1
            tabs = (ptab.to(dev), torch.tensor(heads_list, dtype=torch.int32).to(dev),
                    torch.tensor(offs[:-1], dtype=torch.int64).to(dev), offs[-1], gflat, gt.to(dev))
            _lfs_tables[key] = tabs
        ptab, heads, coef_off, ncoef, gflat, gt = tabs
        offs = [ncoef]
        coef = torch.empty(offs[-1], dtype=torch.float32, device=dev)
        save = torch.empty((nblk, 2, B, 16, 3), dtype=torch.float32, device=dev)
        call('fw_lfs_lambda', xbar, ptab, heads, coef_off, coef, save, nblk, B, C, nb1)
        ctx.save_for_backward(inter, xbar, stats, ptab, heads, coef_off, save, *params)
        ctx.meta = meta
        ctx.gbuf = (gflat, gt)
        ctx.dcoef = torch.zeros_like(coef)          # the attention kernels of all blocks accumulate here (WindowAttnFn dcoef_to)
        return coef

    @staticmethod
    def backward(ctx, dcoef):
        if dcoef is None:
            dcoef = ctx.dcoef
        else:
            dcoef = dcoef + ctx.dcoef
        inter, xbar, stats, ptab, heads, coef_off, save = ctx.saved_tensors[:7]
        params = ctx.saved_tensors[7:]
        heads_list, B, nb1 = ctx.meta
        dev = inter.device
        C = inter.shape[1]
        nblk = len(heads_list)
        gflat, gt = ctx.gbuf
        direct = config.direct_grads and all(p.grad is not None for p in params)
        if direct:
            # engine mode: the kernel's atomics land in the parameters' flat .grad views (704 tiny tensors: no per-tensor add)
            key = ('direct',) + tuple(p.grad.data_ptr() for p in params)
            gt2 = _lfs_tables.get(key)
            if gt2 is None:
                t = torch.zeros((nblk, 2, 8), dtype=torch.int64)
                k = 0
                for bi in range(nblk):
                    for band in range(nb1):
                        for j in range(8):
                            t[bi, band, j] = params[k].grad.data_ptr()
                            k += 1
                gt2 = t.to(dev)
                _lfs_tables[key] = gt2
            gt = gt2
            grads = [None] * len(params)
        else:
            gflat.zero_()
            grads, o = [], 0
            for p in params:
                grads.append(gflat[o:o + p.numel()].view_as(p))
                o += p.numel()
        dxbar = torch.zeros_like(xbar)
        call('fw_lfs_lambda_bwd', xbar, ptab, gt, heads, coef_off, dcoef.contiguous(), save, dxbar, nblk, B, C, nb1)
        dinter = torch.zeros_like(inter)
        call('fw_lfs_xbar_bwd', inter, stats, dxbar, dinter, nb1, B, ctx.NT, C)
        return (dinter, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------------------------
# nn.Dropout as counter-based masks (csrc/fw_common.h: fw_keep): one u32 seed per device, advanced once per training step
# ---------------------------------------------------------------------------------------------------------------
_rng = {}                     # device -> int32 [1] tensor holding the u32 seed
_rng_frozen = [False]


def dropout_seed(device):
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == 'cuda' else 0))
    t = _rng.get(key)
    if t is None:
        if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('fwair: the dropout seed must exist before a stream capture (run one eager step first)')
        v = int(torch.initial_seed()) & 0x7fffffff
        t = _rng[key] = torch.tensor([v], dtype=torch.int32, device=device)
    return t


def set_dropout_seed(value, device, frozen=False):
    """Tests: a known seed; frozen = the per-step tick is disabled, so forward, backward and the oracle see `value`."""
    dropout_seed(device).fill_(int(value) - (1 << 32) if int(value) >= (1 << 31) else int(value))
    _rng_frozen[0] = bool(frozen)


def dropout_tick(device):
    device = torch.device(device)
    if _rng_frozen[0] or device.type != 'cuda':
        return
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key in _rng:
        call('fw_rng_tick', _rng[key])


class DropoutFn(torch.autograd.Function):
    """y = drop(x) on a contiguous f32 tensor (site, p); the backward pass re-derives the mask."""

    @staticmethod
    def forward(ctx, x, site, p):
        x = x.contiguous()
        y = torch.empty_like(x)
        seed = dropout_seed(x.device)
        call('fw_dropout', 0, 0, x, None, None, y, x.numel(), 1, seed, site, p)
        ctx.cfg = (site, p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        site, p, seed = ctx.cfg
        dy = dy.contiguous().float()
        dx = torch.empty_like(dy)
        call('fw_dropout', 0, 0, dy, None, None, dx, dy.numel(), 1, seed, site, p)
        return dx, None, None


class DropAddFn(torch.autograd.Function):
    """y = res + drop(x): a branch's output Dropout followed by the residual add (encoder_ViT.py:33,73,112-116); f32 [M, C]."""

    @staticmethod
    def forward(ctx, x, res, site, p):
        x, res = x.contiguous(), res.contiguous()
        y = torch.empty_like(res)
        seed = dropout_seed(x.device)
        call('fw_dropout', 1, 0, x, None, res, y, x.numel(), 1, seed, site, p)
        ctx.cfg = (site, p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        site, p, seed = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call('fw_dropout', 0, 0, dy, None, None, dx, dy.numel(), 1, seed, site, p)
        return dx, dy, None, None


class GeluDropFn(torch.autograd.Function):
    """g = drop(GELU(h)) on a contiguous T tensor (FeedForward: nn.GELU, nn.Dropout -- encoder_ViT.py:30-31)."""

    @staticmethod
    def forward(ctx, h, site, p):
        assert h.is_contiguous()
        g = torch.empty_like(h)
        seed = dropout_seed(h.device)
        call('fw_dropout', 2, dt(h.dtype), h, None, None, g, h.numel(), 1, seed, site, p)
        ctx.save_for_backward(h)
        ctx.cfg = (site, p, seed)
        return g

    @staticmethod
    def backward(ctx, dg):
        (h,) = ctx.saved_tensors
        site, p, seed = ctx.cfg
        dg = dg.contiguous()
        if dg.dtype != h.dtype:
            dg = dg.to(h.dtype)
        dh = torch.empty_like(h)
        call('fw_dropout', 3, dt(h.dtype), dg, h, None, dh, h.numel(), 1, seed, site, p)
        return dh, None, None


# ---------------------------------------------------------------------------------------------------------------
# DropPath row scales (timm semantics: floor(keep + U) / keep per sample)
# ---------------------------------------------------------------------------------------------------------------
_dp_override = None


def set_droppath_override(fn):
    """fn(name, nsamples, rate, device) -> f32 [nsamples] or None; tests inject recorded masks here."""
    global _dp_override
    _dp_override = fn


class _DropPathPool:
    """All DropPath row scales of one training step from ONE draw.  The calls of a step come in a fixed order; the first complete
    step records (samples, keep) of every call while drawing one by one, later steps draw a single vector at begin() and hand
    out slices -- 3 tiny kernels per step instead of 5 per block.  Any deviation from the recorded plan (another model, batch
    or phase) falls back to per-call draws for the rest of that step and records afresh from the next one."""

    def __init__(self):
        self.plan, self.keep_vec, self.cursor, self.offset, self.pool = [], None, 0, 0, None
        self.state = 'record'                                # 'record' -> 'ready' (plan finalised) ; 'dirty' = give up for this step

    def begin(self, device):
        if self.state == 'record' and self.plan and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            # the previous step recorded a full plan: finalise it (host -> device copy: never while a graph is being captured)
            self.keep_vec = torch.cat([torch.full((n,), k, dtype=torch.float32) for n, k in self.plan]).to(device)
            self.state = 'ready'
        elif self.state == 'dirty' or (self.state == 'ready' and self.cursor != len(self.plan)):
            self.plan, self.state = [], 'record'             # the last step did not follow the plan: record this one
        elif self.state == 'record':
            self.plan = []
        self.cursor = self.offset = 0
        if self.state == 'ready':
            self.pool = torch.floor(self.keep_vec + torch.rand(self.keep_vec.numel(), device=device)) / self.keep_vec

    def draw(self, nsamples, keep, device):
        if self.state == 'ready':
            i = self.cursor
            if i < len(self.plan) and self.plan[i] == (nsamples, keep):
                out = self.pool[self.offset:self.offset + nsamples]
                self.cursor, self.offset = i + 1, self.offset + nsamples
                return out
            self.state = 'dirty'
        elif self.state == 'record':
            self.plan.append((nsamples, keep))
        return torch.floor(keep + torch.rand(nsamples, device=device)) / keep


_dp_pools = {}
_dp_current = [None]


def droppath_begin(device, key='step'):
    """Call once at the start of every training forward (net.model does).  key: one plan per kind of step (full model /
    encoder only), so alternating phases do not invalidate each other's plan."""
    pool = _dp_pools.get(key)
    if pool is None:
        pool = _dp_pools[key] = _DropPathPool()
    _dp_current[0] = pool
    pool.begin(device)
    dropout_tick(device)                                   # a new training step: new Dropout masks (no-op without a Dropout user)


def droppath_scale(name, nsamples, rate, training, device):
    if not training or rate == 0.0:
        return None
    if _dp_override is not None:
        return _dp_override(name, nsamples, rate, device)
    if _dp_current[0] is None:
        droppath_begin(device)
    return _dp_current[0].draw(nsamples, 1.0 - rate, device)
