"""Tensor-level wrappers over the C ABI (include/fwair.h).  They only allocate outputs with the PyTorch
caching allocator, check shapes on the host and forward raw pointers; all arithmetic happens in
libfwair_hip.so.  Activations are 2-D [tokens, channels] tensors whose row stride is their `ld`."""
import torch

from . import lfs as _lfs
from .lib import call, dt, lib


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, 'expect a row-major 2-D view'
    return t.stride(0)


def empty(rows, cols, dtype, device):
    return torch.empty((rows, cols), dtype=dtype, device=device)


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(x, w, M, N, K, *, x_trans=False, w_trans=False, x_op=0, w_op=0, out=None, out_dtype=None, bias=None,
         act=0, slope=0.0, aux=None, rowscale=None, rows_per_scale=1, residual=None, accumulate=False, splitk=1,
         alpha=1.0, out_gelu=None, xsum=None, c_zstride=0, xsum_zstride=0):
    """C[M,N] = epi(alpha * X W^T).  x: [M,K] (or [K,M] when x_trans), w: [N,K] (or [K,N] when w_trans)."""
    assert x.dtype == w.dtype
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or x.dtype, device=x.device)
    out_f32 = int(out.dtype == torch.float32)
    call('fw_gemm', dt(x.dtype), x, _ld(x), int(x_trans), x_op, w, _ld(w), int(w_trans), w_op, out, _ld(out), out_f32,
         int(accumulate), M, N, K, float(alpha), bias, act, float(slope), aux, _ld(aux) if aux is not None else 0,
         rowscale, rows_per_scale, residual, _ld(residual) if residual is not None else 0, splitk,
         out_gelu, _ld(out_gelu) if out_gelu is not None else 0, xsum, c_zstride, xsum_zstride)
    return out


import os as _os
_SPLITK_BLOCKS = int(_os.environ.get('FW_SPLITK_BLOCKS', '512'))       # tuning knob of pick_splitk (blocks in flight aimed at)
_DGRAD_SPLIT_ROWS = int(_os.environ.get('FW_DGRAD_SPLIT_ROWS', '512'))   # reduction rows per slice of a split input gradient (ops.dgrad)


def pick_splitk(M, N, K, dtype):
    """Split factor of a weight-gradient GEMM (small M x N output, long reduction K = tokens).  Rule fitted to sweeps on MI355X
    (tools/splitk_sweep.py, tools/wgrad_probe.py): as many blocks as fit ONE round of the chip (256 CUs x 2 workgroups = 512) and
    never more -- 528 blocks ran 77 us where 440 ran 56 (a second round for 16 stragglers) --, with at least 512 reduction rows
    (8 K-steps) per block.  (fw_gemm deals contiguous shares of the tile order to the 8 XCDs for ANY block count.)"""
    tiles = ((M + 127) // 128) * ((N + 63) // 64 if N <= 64 else (N + 127) // 128)
    sk = max(1, min(_SPLITK_BLOCKS // max(tiles, 1), K // 512))
    while sk > 1 and sk * M * N * 4 > (256 << 20):          # keep the partial-tile slab under 256 MB
        sk //= 2
    return int(sk)


# ---- deferred reduction of split partials -------------------------------------------------------------------
# A backward pass produces several hundred small slabs of partial sums (split weight-gradient GEMMs, LayerNorm column sums)
# whose results nobody reads before the optimizer step.  Inside a backward pass they are queued and folded by ONE
# fw_slab_reduce_multi launch when the pass ends (autograd engine callback) instead of one tiny launch each.
_pending = []            # (slab, nz, n, zstride, dst, dst2, off2, n2)
_flush_registered = [None]      # graph-task id of the backward pass whose end-of-pass callback is queued
_keepalive = []          # pinned host tables referenced by captured HIP graphs (their memcpy nodes re-read them on replay)


_HOST_WORDS = 65536      # int64 words per pinned table: room for ~1400 slabs, or ~28 000 work items of a grouped weight-gradient launch
_host_ring, _host_next, _host_reserved = [], [0], []


def reserve_capture_tables(count=None):
    """Pinned host memory cannot be allocated while a stream is being captured: the engine reserves the tables a capture
    will consume beforehand.  A captured table is never reused (the graph's memcpy node re-reads it at every replay)."""
    if count is None:                                      # one per grouped weight-gradient launch and per slab fold of a captured pass
        count = 6
    while len(_host_reserved) < count:
        _host_reserved.append(torch.empty((_HOST_WORDS,), dtype=torch.int64, pin_memory=True))


def _host_table(words):
    assert words <= _HOST_WORDS, 'too many pending slabs for one table'
    if torch.cuda.is_current_stream_capturing():
        if not _host_reserved:
            raise RuntimeError('fwair.ops: call reserve_capture_tables() before capturing a backward pass')
        buf = _host_reserved.pop()
        _keepalive.append(buf)
        return buf[:words]
    if len(_host_ring) < 4:
        _host_ring.append([torch.empty((_HOST_WORDS,), dtype=torch.int64, pin_memory=True), None])
    slot = _host_ring[_host_next[0] % len(_host_ring)]
    _host_next[0] += 1
    if slot[1] is not None:
        slot[1].synchronize()                          # the async copy that last read this buffer has finished
    slot[1] = torch.cuda.Event()
    return slot[0][:words]


def _in_backward():
    return torch._C._current_graph_task_id() != -1


def slab_reduce(slab, nz, n, zstride, dst, dst2=None, off2=0, n2=0, defer=False):
    """dst[0:n] += sum_z slab[z][0:n];  dst2[0:n2] += sum_z slab[z][off2:off2+n2].  With defer (the destinations are persistent
    gradient buffers, not tensors handed back to autograd) the fold waits for the end of the running backward pass."""
    if not defer or not _in_backward():
        call('fw_slab_reduce', slab, nz, n, zstride, dst, 1, dst2, off2, n2 if dst2 is not None else 0)
        return
    _register_flush()
    _pending.append((slab, nz, n, zstride, dst, dst2, off2, n2 if dst2 is not None else 0))


def flush_slabs():
    """End-of-backward-pass callback: the grouped weight gradients first (they queue the slabs of their sliced reductions), then ONE
    fold of every slab of the pass."""
    _flush_registered[0] = None
    _flush_wgrads()
    _fold_slabs()


def _fold_slabs():
    if not _pending:
        return
    items = list(_pending)
    _pending.clear()
    num = len(items)
    dev = items[0][0].device
    host = _host_table(num * 12 + num + 1)
    tab, prefix = host[:num * 12].view(num, 12), host[num * 12:]
    rows, offs, total = [], [0], 0
    seen = {}
    for it in items:                                           # destinations written by more than one entry need atomics
        for t in (it[4], it[5]):
            if t is not None:
                seen[t.data_ptr()] = seen.get(t.data_ptr(), 0) + 1
    for slab, nz, n, zstride, dst, dst2, off2, n2 in items:
        end = off2 + n2 if dst2 is not None else n
        nchunks = ((end + 3) // 4 + 63) // 64                  # 64 lanes x 16 bytes
        zper = min(nz, 64)
        splits = (nz + zper - 1) // zper
        upw = max(1, 32 // zper)                               # units per wave: >= ~32 KB moved by every wave
        blocks = (nchunks * splits + 4 * upw - 1) // (4 * upw)
        shared = seen[dst.data_ptr()] > 1 or (dst2 is not None and seen[dst2.data_ptr()] > 1)
        unaligned = dst.data_ptr() % 16 or (dst2 is not None and dst2.data_ptr() % 16)
        rows.append((slab.data_ptr(), dst.data_ptr(), dst2.data_ptr() if dst2 is not None else 0, n, zstride, off2, n2, nz, zper,
                     nchunks, upw, int(splits > 1 or shared or bool(unaligned))))
        total += blocks
        offs.append(total)
    tab.copy_(torch.tensor(rows, dtype=torch.int64))
    prefix.copy_(torch.tensor(offs, dtype=torch.int64))
    table = torch.empty(host.shape, dtype=torch.int64, device=dev)
    table.copy_(host, non_blocking=True)
    call('fw_slab_reduce_multi', table, table[num * 12:], num, total)
    if not torch.cuda.is_current_stream_capturing():
        for slot in _host_ring:
            if slot[0].data_ptr() == host.data_ptr():
                slot[1].record()
    # `items` (the slabs) die here: the allocator reuses them stream-ordered, i.e. after the kernel above


# ---- grouped weight gradients: every dW = dY^T x of a backward pass in one launch (csrc/fw_gemm.hip: gemm_wgrad_group_kernel) --------
_GROUP = int(_os.environ.get('FW_WGRAD_GROUP', '1'))                # 0: one launch per product (round-2 behaviour)
_GROUP_CHUNK = int(_os.environ.get('FW_WGRAD_CHUNK', '4096'))       # tokens per work item: longer reductions are cut into slices (sweep on MI355X: 1024: 298.6, 2048: 307.5, 4096: 305.6-308.7, 8192: 302.3, 16384: 281.3 images/s)
_GROUP_BIG_MIN = int(_os.environ.get('FW_WGRAD_BIG_MIN', '224'))  # smallest output side that takes the 256 x 256 tile form
_GROUP_PLAIN = int(_os.environ.get('FW_WGRAD_PLAIN', '1'))        # sole writer of a (pre-zeroed) gradient: plain store instead of accumulate
_GROUP_UNIT_BX = int(_os.environ.get('FW_WGRAD_UNIT_BX', '1'))    # dY column blocks per work unit (see _launch_group)
_pending_w = []          # (g, x, n, k, m, dw, db)
_shared_dw = set()
_fresh_dw = set()        # gradients the caller vouches are ZERO when the pass starts (engine mode) and that ONE product writes: plain store


def _register_flush():
    task = torch._C._current_graph_task_id()
    if _flush_registered[0] != task:                       # a new backward pass (an earlier one may have died before its callback ran)
        if _flush_registered[0] is not None:
            _pending.clear(); _pending_w.clear(); _fresh_dw.clear()      # work of a pass that never finished: its gradients are void anyway
        _flush_registered[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(flush_slabs)


def _groupable(g, x, n, k, m, dw, db):
    return (_GROUP and g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and m % 32 == 0 and g.stride(1) == 1 and x.stride(1) == 1
            and g.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and g.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and n >= 8 and k % 4 == 0
            and dw.dtype == torch.float32 and dw.stride(1) == 1 and dw.stride(0) % 4 == 0 and dw.data_ptr() % 16 == 0)


def _flush_wgrads():
    """Launch the queued weight gradients: ONE fw_wgrad_group over (problem, tile, slice) work items.  A product whose reduction is
    longer than FW_WGRAD_CHUNK tokens is cut into slices that store partial tiles into a slab (folded with the other slabs of the
    pass right after); everything else owns its whole reduction and adds straight into the gradient -- no slab at all."""
    if not _pending_w:
        return
    work = list(_pending_w)
    _pending_w.clear()
    seen = set()
    _shared_dw.clear()
    for w_ in work:                                          # gradients that several products of this pass add into keep the accumulate form
        (_shared_dw if w_[5].data_ptr() in seen else seen).add(w_[5].data_ptr())
    _shared_dw.update(seen - _fresh_dw)                      # ... and so does everything nobody declared fresh
    _fresh_dw.clear()
    # outputs of at least _GROUP_BIG_MIN rows and columns run on 256 x 256 tiles (8 waves), the others on 128 x 128 (4 waves)
    big = [w for w in work if min(w[2], w[3]) >= _GROUP_BIG_MIN]
    small = [w for w in work if min(w[2], w[3]) < _GROUP_BIG_MIN]
    if big:
        _launch_group(big, 256)
    if small:
        _launch_group(small, 128)


def _group_fire(table, probs, nprob, total, tile):
    """The launch itself, apart from the host-side list building (bench.py times exactly this call)."""
    call('fw_wgrad_group', table, probs, nprob, table[nprob * 16:], total, tile)


def _launch_group(work, tile=128):
    dev = work[0][0].device
    rows, units = [], []
    for pi, (g, x, n, k, m, dw, db) in enumerate(work):
        tm, tn = (n + tile - 1) // tile, (k + tile - 1) // tile
        sk = max(1, -(-m // _GROUP_CHUNK))
        kper = -(-(-(-m // sk)) // 32) * 32                   # tokens per slice, whole 32-token steps
        sk = -(-m // kper)
        if sk == 1:
            sole = dw.data_ptr() not in _shared_dw                      # fresh, and the only product of the pass that writes it
            rows.append((g.data_ptr(), x.data_ptr(), dw.data_ptr(), g.stride(0), x.stride(0), dw.stride(0), n, k, m, kper, 1,
                         db.data_ptr() if db is not None else 0, 1 if sole else 0, 0, 0 if sole else 1, 0))   # c_zstride > 0 selects the plain-store tile
        else:
            nk = (n * k + 3) // 4 * 4
            S = nk + (n + 3) // 4 * 4
            slab = torch.empty((sk, S), dtype=torch.float32, device=dev)
            rows.append((g.data_ptr(), x.data_ptr(), slab.data_ptr(), g.stride(0), x.stride(0), k, n, k, m, kper, sk,
                         slab.data_ptr() + 4 * nk if db is not None else 0, S, S if db is not None else 0, 0, 0))
            assert dw.is_contiguous() or dw.stride(0) == k
            _pending.append((slab, sk, n * k, S, dw, db, nk, n if db is not None else 0))
        for z in range(sk):
            # unit = a block of _GROUP_UNIT_BX x tn tiles of one slice, run back to back on ONE XCD: its tiles share dY column blocks
            # (same bx) and x column blocks (same by) in that L2 -- HBM reads per unit ~ (u + tn) operand blocks for u * tn tiles
            for bx in range(0, tm, _GROUP_UNIT_BX):
                units.append((min(kper, m - z * kper), pi, bx, min(_GROUP_UNIT_BX, tm - bx), tn, z))
    # Workgroups are dealt round-robin to the 8 XCDs (block p runs on XCD p % 8, in block order).  Units are handed, longest first, to the
    # XCD with the least work so far (every XCD then runs ITS list longest-first: a long tile started last would be the tail); lists are
    # padded to one length with empty items (problem -1).
    units.sort(key=lambda u: -u[0])
    lists, load = [[] for _ in range(8)], [0] * 8
    for ln, pi, bx, u, tn, z in units:
        x = load.index(min(load))
        load[x] += ln * tn * u
        lists[x].extend((pi, bx + i, by, z) for by in range(tn) for i in range(u))
    depth = max(len(li) for li in lists)
    total = depth * 8
    items = [lists[p & 7][p >> 3] if (p >> 3) < len(lists[p & 7]) else (-1, 0, 0, 0) for p in range(total)]
    nprob = len(rows)
    words = nprob * 16 + total * 2                               # items: 4 int32 = 2 int64 words each
    if words > _HOST_WORDS and len(work) > 1:                    # more work than one pinned table describes: two launches
        del _pending[len(_pending) - sum(1 for r_ in rows if r_[10] > 1):]          # the slabs queued above are re-made by the halves
        _launch_group(work[:len(work) // 2], tile); _launch_group(work[len(work) // 2:], tile)
        return
    host = _host_table(words)
    host[:nprob * 16].view(nprob, 16).copy_(torch.tensor(rows, dtype=torch.int64))
    it32 = torch.tensor(items, dtype=torch.int32)
    host[nprob * 16:].view(torch.int32).view(total, 4).copy_(it32)
    table = torch.empty(host.shape, dtype=torch.int64, device=dev)
    table.copy_(host, non_blocking=True)
    probs = torch.empty(nprob * lib().fw_wgrad_group_prob_bytes(), dtype=torch.uint8, device=dev)
    _group_fire(table, probs, nprob, total, tile)
    if not torch.cuda.is_current_stream_capturing():
        for slot in _host_ring:
            if slot[0].data_ptr() == host.data_ptr():
                slot[1].record()
    # `work` (the operands) dies here: the allocator reuses them stream-ordered, i.e. after the kernel above


def wgrad(g, x, n, k, m, dw, db=None, defer=False, fresh=False):
    """dw[n][k] += sum_m g[m][n] x[m][k];  db[n] += sum_m g[m][n].  With defer (dw / db are persistent gradient buffers nobody reads
    before the optimizer step) inside a backward pass the product is QUEUED and runs in the pass's grouped launch (_flush_wgrads).
    fresh: the caller vouches that dw / db are ZERO when the pass starts (the engine's flat gradient buffer): if this is the only queued
    product of the pass that writes them, its tiles are STORED instead of read-modified-written.
    Otherwise: large reductions are split over K into a slab of partial tiles (plain stores) that a slab reduce folds -- no
    same-address atomics."""
    if defer and _in_backward() and _groupable(g, x, n, k, m, dw, db):
        _register_flush()
        _pending_w.append((g, x, n, k, m, dw, db))
        if fresh and _GROUP_PLAIN:
            _fresh_dw.add(dw.data_ptr())
        return
    sk = pick_splitk(n, k, m, g.dtype)
    if sk == 1:
        gemm(g, x, n, k, m, x_trans=True, w_trans=True, out=dw, accumulate=True, xsum=db)
        return
    nk = (n * k + 3) // 4 * 4
    S = nk + (n + 3) // 4 * 4
    slab = torch.empty((sk, S), dtype=torch.float32, device=g.device)
    cview = slab[0, :n * k].view(n, k)
    xs = slab[0, nk:] if db is not None else None
    gemm(g, x, n, k, m, x_trans=True, w_trans=True, out=cview, splitk=sk, xsum=xs, c_zstride=S, xsum_zstride=S if db is not None else 0)
    slab_reduce(slab, sk, n * k, S, dw, db, nk, n, defer=defer)


def dgrad(g, w, M, K, N, out, act=0, aux=None):
    """out[M,K] = epi(g[M,N] W[N,K])  (input gradient of a Linear).  A long reduction into a small output -- the 65536-wide
    mlp_head of the encoder: 32 output tiles, 1024 K-steps each -- is split over N into a slab of partial tiles."""
    tiles = ((M + 127) // 128) * ((K + 127) // 128)
    # fewer than 200 output tiles leave the chip to a handful of long K loops (1024 x 896 x 3584: 56 tiles, 56 steps each, 63 us):
    # split the reduction so that one round of the chip is busy, at >= _DGRAD_SPLIT_ROWS reduction rows per slice
    sk = min(512 // max(tiles, 1), N // _DGRAD_SPLIT_ROWS) if (act == 0 and tiles < 200) else 1
    if sk <= 1:
        gemm(g, w, M, K, N, w_trans=True, out=out, act=act, aux=aux)
        return
    S = (M * K + 3) // 4 * 4
    slab = torch.empty((sk, S), dtype=torch.float32, device=g.device)
    gemm(g, w, M, K, N, w_trans=True, out=slab[0, :M * K].view(M, K), splitk=sk, c_zstride=S)
    direct = out.dtype == torch.float32 and out.is_contiguous()
    tmp = out if direct else torch.empty((M, K), dtype=torch.float32, device=g.device)
    call('fw_slab_reduce', slab, sk, M * K, S, tmp, 0, None, 0, 0)
    if not direct:
        call('fw_cast_rows', dt(out.dtype), tmp, K, out, _ld(out), M, K, None, 1)


# ------------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, out_dtype, eps=1e-5):
    rows, C = x.shape
    y = torch.empty((rows, C), dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call('fw_layernorm_fwd', dt(out_dtype), x, _ld(x), gamma, beta, y, _ld(y), mean, rstd, rows, C, eps)
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, defer=False, twin=None, twscale=None, tw_rows_per_scale=1):
    """twin: optional T [rows, C] buffer that also receives dx * twscale[row // tw_rows_per_scale] (see fw_layernorm_bwd2)."""
    rows, C = x.shape
    dx = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    nblk = lib().fw_layernorm_bwd_blocks(rows, C)
    partial = torch.empty((nblk, 2 * C), dtype=torch.float32, device=x.device)
    call('fw_layernorm_bwd2', dt(dy.dtype), dy, _ld(dy), x, _ld(x), gamma, mean, rstd, dres,
         _ld(dres) if dres is not None else 0, dx, _ld(dx), None, None, partial, rows, C,
         twin, _ld(twin) if twin is not None else 0, twscale, tw_rows_per_scale)
    slab_reduce(partial, nblk, C, 2 * C, dgamma, dbeta, C, C, defer=defer)
    return dx


# ------------------------------------------------------------------------------------------------ attention
def attn_fwd(qkv, C, B, H, W, heads, L, mode, shift, bias, coef=None, lfs=0):
    """qkv: [L*B*H*W, 3C] (q | k | v).  bias: f32 [L*L or 1, 225, heads].  -> out [rows, C], lse."""
    rows = qkv.shape[0]
    D = C // heads
    nkt = 1 if mode == 0 else L - 1
    out = torch.empty((rows, C), dtype=qkv.dtype, device=qkv.device)
    nitems = B * (H // 8) * (W // 8) * L * heads
    lse = torch.empty((nitems, 64), dtype=torch.float32, device=qkv.device)
    tab = _lfs.device_table(qkv.dtype, qkv.device) if lfs == 2 else None
    call('fw_attn_fwd', dt(qkv.dtype), D, nkt, lfs, qkv, qkv[:, C:], qkv[:, 2 * C:], _ld(qkv), out, _ld(out), lse, bias,
         coef, tab, B, H, W, heads, L, mode, shift, float(D) ** -0.5)
    return out, lse


def attn_bwd(qkv, out, dout, lse, C, B, H, W, heads, L, mode, shift, bias, dbias_dense, coef=None, dcoef=None, lfs=0):
    """-> dqkv [rows, 3C].  dbias_dense f32 [ntab, heads, 64, 64] and dcoef are accumulated into."""
    rows = qkv.shape[0]
    D = C // heads
    nkt = 1 if mode == 0 else L - 1
    dqkv = torch.empty_like(qkv)
    d2 = torch.empty_like(qkv) if nkt == 2 else None          # second key-gradient slot, same layout as dqkv
    tab = _lfs.device_table(qkv.dtype, qkv.device) if lfs == 2 else None
    call('fw_attn_bwd', dt(qkv.dtype), D, nkt, lfs, qkv, qkv[:, C:], qkv[:, 2 * C:], _ld(qkv), out, _ld(out), dout, _ld(dout),
         lse, bias, coef, tab, dqkv, dqkv[:, C:], dqkv[:, 2 * C:], d2[:, C:] if d2 is not None else None,
         d2[:, 2 * C:] if d2 is not None else None, _ld(dqkv),
         dbias_dense, dcoef, B, H, W, heads, L, mode, shift, float(D) ** -0.5, 0)
    if nkt == 2:
        call('fw_add_rows', dt(qkv.dtype), d2[:, C:], _ld(d2), dqkv[:, C:], _ld(dqkv), rows, 2 * C)
    return dqkv


def rel_index(device):
    """[64*64] int64 index into the 225-entry table (decoder_Uformer.py:201-210), host-built constant."""
    ch = torch.arange(8)
    coords = torch.stack(torch.meshgrid([ch, ch], indexing='ij')).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += 7
    rel[:, :, 1] += 7
    rel[:, :, 0] *= 15
    return rel.sum(-1).to(device)


# ------------------------------------------------------------------------------------------------ LeFF dwconv
def dwconv_fwd(src, w, bias, B, H, W, in_gelu=False):
    """w: f32 tap-major [9, C].  src = g1, or (in_gelu) the pre-activation h1.  -> (h2, g2 = GELU(h2))"""
    h2, g2 = torch.empty_like(src), torch.empty_like(src)
    call('fw_dwconv_fwd', dt(src.dtype), src, _ld(src), int(in_gelu), w, bias, h2, g2, _ld(h2), B, H, W, src.shape[1])
    return h2, g2


def dwconv_bwd(dh2, g1, h1, w, dw, dbias, B, H, W):
    """g1 may be None: the weight gradient then evaluates GELU(h1) itself."""
    dh1 = torch.empty_like(h1)
    call('fw_dwconv_bwd', dt(h1.dtype), dh2, _ld(dh2), g1, h1, _ld(h1), w, dh1, _ld(dh1), dw, dbias, B, H, W, h1.shape[1])
    return dh1


# ------------------------------------------------------------------------------------------------ convs
def im2col4(x, B, H, W, dtype):
    C = x.shape[1]
    col = torch.empty((B * (H // 2) * (W // 2), 16 * C), dtype=dtype, device=x.device)
    call('fw_im2col4', dt(dtype), x, _ld(x), col, B, H, W, C)
    return col


def col2im4(dcol, B, H, W, C, dres=None):
    dx = torch.empty((B * H * W, C), dtype=torch.float32, device=dcol.device)
    call('fw_col2im4', dt(dcol.dtype), dcol, dx, _ld(dx), dres, _ld(dres) if dres is not None else 0, B, H, W, C)
    return dx


def pixel_shuffle(g, bias, out, B, H, W, Cout):
    call('fw_pixel_shuffle', dt(g.dtype), g, bias, out, _ld(out), B, H, W, Cout)
    return out


def pixel_unshuffle(dout, B, H, W, Cout, dtype):
    dg = torch.empty((B * H * W, 4 * Cout), dtype=dtype, device=dout.device)
    call('fw_pixel_unshuffle', dt(dtype), dout, _ld(dout), dg, B, H, W, Cout)
    return dg


def colsum(x, out):
    call('fw_colsum', 1 if x.dtype == torch.bfloat16 else 0, x, _ld(x), out, x.shape[0], x.shape[1])


def cast_rows(src, dtype, rowscale=None, rows_per_scale=1):
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    call('fw_cast_rows', dt(dtype), src, _ld(src), dst, _ld(dst), src.shape[0], src.shape[1], rowscale, rows_per_scale)
    return dst


def copy_rows(src, dst, accumulate=False):
    call('fw_copy_rows', src, _ld(src), dst, _ld(dst), src.shape[0], src.shape[1], int(accumulate))


def permute3(src, dst, dims, out_strides, accumulate=False):
    ind = 1 if src.dtype == torch.bfloat16 else 0
    outd = 1 if dst.dtype == torch.bfloat16 else 0
    call('fw_permute3', ind, outd, src, dst, dims[0], dims[1], dims[2], out_strides[0], out_strides[1], out_strides[2],
         int(accumulate))
