#!/usr/bin/env python3
"""bench.py -- training images/sec of the AirNet hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype bf16|fp32] [--no-graph] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one phase-2 training step of train.py:80-96 on one synthetic batch: zero grads, AirNet forward
(query encoder, key-encoder EMA + forward, MoCo logits, decoder with learned frequency selection), L1 + 0.6 * CE,
backward, (gradient all-reduce over RCCL), Adam.  Workload = BASELINE.json configs[1]: Uformer encoder + decoder,
denoise sigma = 25, 128x128, batch 16 per GPU, bf16 operands / f32 accumulation.  Weak scaling (fixed per-GPU batch).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd')
sys.path.insert(0, PKG)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL across processes needs it on this driver

import numpy as np          # noqa: E402
import torch                # noqa: E402

FLOP_PER_IMAGE_STEP = 554.9e9        # 3*(34.72+138.66)+34.72 GFLOP, dense contractions only
PEAK_BF16 = 2.5e15                   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA = 157.3e12
PEAK_HBM = 8.0e12
PEAK_FOR = {'bf16': PEAK_BF16, 'f32': PEAK_F32_MFMA}


from fwair.synthetic import synth_batch      # noqa: E402  (SURVEY.md 8(d) generator, shared with train_ddp.py)


def make_opt(batch, dtype, patch=128, encoder='Uformer', tasks=None):
    import types
    o = types.SimpleNamespace(L=3, encoder_dim=256, encoder_embed_dim=28, embed_dim=56, batch_size=batch, patch_size=patch,
                              degradation_embedding_method=['all_3_bands'], encoder_msa_type='freq', contrast_loss_weight=0.6,
                              encoder_type='Uformer', decoder_type='Uformer', debug_mode=False, frequency_decompose_type='none',
                              learnable_modulator=False, compute_dtype=dtype, de_type=[(tasks or ['denoising_25'])[i % len(tasks or [1])]
                                                                                        for i in range(batch)])
    if encoder == 'ViT':          # BASELINE configs[4]: ViT encoder (N = (patch/16)^2 tokens) + plain Uformer decoder (option.py:80-101: encoder_dim 3)
        o.encoder_type, o.encoder_dim, o.degradation_embedding_method, o.out_channels, o.batch_wise_decompose = 'ViT', 3, ['None'], 3, False
    return o


def flop_per_image_step(encoder, patch):
    """Dense-contraction FLOPs of one training step per image: 3 x (query encoder + decoder) + key-encoder forward."""
    px = (patch / 128) ** 2
    dec = 138.66e9 * px
    if encoder == 'ViT':
        n = (patch // 16) ** 2
        enc = 12 * (2 * n * (768 * 2304 + 768 * 768 + 2 * 768 * 3072) + 4 * n * n * 768) + 2 * n * 768 * 768 + 2 * n * 768 * 768
    else:
        enc = 34.72e9 * px
    return 3 * (enc + dec) + enc


def gemm_profile(engine, batch, reps=8):
    """Per-launch GPU time of every distinct fw_gemm call of one training step.

    Pass 1 runs one eager step with a hook that records the signature of each launch (shapes, strides, dtypes, epilogue
    flags).  Pass 2 re-creates operands of each DISTINCT signature and times `reps` back-to-back launches inside a captured
    HIP graph with HIP events on the capture stream -- the same launch path the timed region uses, so the numbers are
    kernel durations (they agree with rocprofv3's per-kernel averages), free of host launch gaps.
    Per launch: algorithmic FLOPs 2*M*N*K and algorithmic bytes = every operand / result element touched once."""
    from fwair import ops
    rec = []
    orig = ops.gemm

    def spec(t):
        return None if t is None else (tuple(t.shape), tuple(t.stride()), t.dtype)

    def hook(x, w, M, N, K, **kw):
        r = orig(x, w, M, N, K, **kw)
        tens = {k: spec(v) for k, v in kw.items() if isinstance(v, torch.Tensor)}
        tens['x'], tens['w'] = spec(x), spec(w)
        if 'out' not in tens:
            tens['out'] = spec(r)
        scal = tuple(sorted((k, v) for k, v in kw.items() if not isinstance(v, torch.Tensor) and v is not None and k != 'out_dtype'))
        rec.append(((M, N, K), tuple(sorted(tens.items())), scal))
        return r

    # the weight gradients of a backward pass do not come through ops.gemm: they are queued and run as ONE grouped launch per tile
    # form (ops._launch_group -> fw_wgrad_group).  Those launches last milliseconds, so HIP events around the launch call (ops._group_fire: the
    # host-side list building stays outside) inside eager steps time them (mean of 3 steps); their FLOPs / bytes are the sums over the queued products.
    grp, cur, orig_lg, orig_fire = [], [], ops._launch_group, ops._group_fire

    def lg_hook(work, tile=128):
        cur.append([(n, k, m, g.element_size(), dw.element_size(), db is not None) for (g, x, n, k, m, dw, db) in work])
        try:
            orig_lg(work, tile)                          # a table overflow re-enters with halves: each half is its own launch and row
        finally:
            cur.pop()

    def fire_hook(table, probs, nprob, total, tile):     # the launch alone: the host-side list building stays outside the events
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_fire(table, probs, nprob, total, tile)
        e1.record()
        grp.append((tile, cur[-1], e0, e1))

    ops.gemm = hook
    try:
        engine.step_eager(*batch)
        torch.cuda.synchronize()
        ops.gemm = orig
        ops._launch_group, ops._group_fire = lg_hook, fire_hook
        GROUP_STEPS = 3
        for _ in range(GROUP_STEPS):
            engine.step_eager(*batch)
        torch.cuda.synchronize()
    finally:
        ops.gemm = orig
        ops._launch_group, ops._group_fire = orig_lg, orig_fire
    counts = {}
    for sig in rec:
        counts[sig] = counts.get(sig, 0) + 1

    def make(sp):
        shape, stride, dtype = sp
        n = 1 + sum((d - 1) * st for d, st in zip(shape, stride))
        base = (torch.randn(n, device='cuda') * 0.5).to(dtype)
        return torch.as_strided(base, shape, stride)

    side = torch.cuda.Stream()
    from fwair.lib import lib as _lib
    L = _lib()
    out = []
    for (M, N, K), tens, scal in counts:
        cnt = counts[((M, N, K), tens, scal)]

        def operands():
            t = {k: make(v) for k, v in tens}
            kw = dict(scal)
            x, w = t.pop('x'), t.pop('w')
            kw.update(t)
            zs = int(kw.get('c_zstride', 0) or 0)
            if zs > 0:                                   # split-K slab (ops.wgrad / ops.dgrad): every z slice must exist
                sk_ = int(kw['splitk'])
                slab = torch.zeros((sk_ + 1) * zs, dtype=torch.float32, device='cuda')
                kw['out'] = slab[:M * N].view(M, N)
                if kw.get('xsum') is not None:
                    nk = (M * N + 3) // 4 * 4
                    assert int(kw.get('xsum_zstride', 0)) == zs and nk + M <= zs
                    kw['xsum'] = slab[nk:zs]
            return x, w, kw

        # every repetition gets its OWN operands: in the step no launch finds its inputs in the 256 MB infinity cache, and
        # eight launches over one buffer set would (they measured 14 % shorter than the rocprofv3 in-step average)
        sets = [operands() for _ in range(reps)]
        x, w, kw = sets[0]
        with torch.cuda.stream(side):
            orig(x, w, M, N, K, **kw)
            nbuf = ctypes.create_string_buffer(96)
            L.fw_gemm_last_kernel(nbuf, 96)                   # the kernel fw_gemm really chose (named as in the rocprofv3 stats)
            kern = nbuf.value.decode()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for xs, ws, kws in sets:
                    orig(xs, ws, M, N, K, **kws)
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            g.replay()
            e1.record(side)
            torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / reps
        sz = x.element_size()
        sk = max(1, int(kw.get('splitk', 1) or 1))
        # STRICT algorithmic bytes: every operand and the result touched ONCE -- X + W + C (+ the fused epilogue's own
        # operands: residual / GELU' argument read, GELU twin written).  The implementation's split-K partial slabs
        # ((sk - 1) extra copies of C, re-read by the slab fold) and the read-modify-write of an accumulating C are
        # OVERHEAD, tallied separately and never credited to `achieved`.
        by = (M * K + N * K) * sz + M * N * kw['out'].element_size()
        over = M * N * kw['out'].element_size() * (2 * sk - 1 if sk > 1 else 0)   # (sk-1) extra slab writes + sk slab reads of the fold
        if kw.get('accumulate'):
            over += M * N * kw['out'].element_size()
        for key in ('residual', 'aux', 'out_gelu'):
            if kw.get(key) is not None:
                by += M * N * kw[key].element_size()
        variant = ('bf16' if x.dtype == torch.bfloat16 else 'f32', kern)
        flags = '+'.join([f'{k}={v}' for k, v in scal if k in ('act', 'accumulate', 'alpha', 'x_op', 'w_op', 'rows_per_scale')] +
                         [k for k, _ in tens if k not in ('x', 'w', 'out')] + [str(kw['out'].dtype).replace('torch.', 'out_')])
        out.append((variant, (M, N, K, sk, flags), cnt, 2.0 * M * N * K, float(by), dt, float(over)))
        del g, sets, kw, x, w
    chunk = ops._GROUP_CHUNK
    by_tile = {}
    for tile, probs, e0, e1 in grp:
        by_tile.setdefault(tile, []).append((probs, e0.elapsed_time(e1) * 1e-3))
    for tile, runs in by_tile.items():
        per_step = len(runs) // GROUP_STEPS                   # launches of this tile form in one step
        dt = sum(t for _, t in runs) / len(runs)
        fl = by = over = 0.0
        for probs, _ in runs[:per_step]:
            for n, k, m, sz, osz, has_b in probs:
                fl += 2.0 * n * k * m
                by += (n * m + k * m) * sz + n * k * osz + (n * osz if has_b else 0)
                sk = max(1, -(-m // chunk))
                over += n * k * osz * ((2 * sk - 1) if sk > 1 else (0 if ops._GROUP_PLAIN else 1))   # sliced: slab writes + fold reads; whole: the RMW read of dW unless stored plainly (sole writer of a pre-zeroed gradient)
        nprob = sum(len(p) for p, _ in runs[:per_step])
        kern = 'gemm_wgrad_group_big_kernel' if tile == 256 else 'gemm_wgrad_group_kernel'
        # one row per tile form: "launch" = the mean of its launches of a step; FLOPs / bytes per launch likewise
        out.append((('bf16', kern), (nprob, 0, 0, 1, f'grouped dW=dY^T x: {nprob} products in {per_step} launch(es) of {tile}x{tile} tiles'),
                    per_step, fl / per_step, by / per_step, dt, over / per_step))
    agg = {}
    for variant, shape, cnt, fl, by, dt, over in out:
        d = agg.setdefault(variant, [0.0, 0.0, 0, 0.0, 0.0, 0.0])
        d[0] += fl * cnt; d[1] += dt * cnt; d[2] += cnt; d[3] += by * cnt
        d[4] += cnt * max(fl / PEAK_FOR[variant[0]], by / PEAK_HBM)          # time the launches would take at their roofline
        d[5] += over * cnt
    dump = os.environ.get('FW_GEMM_DUMP')
    if dump:
        with open(dump, 'w') as f:
            f.write('dtype,kernel,M,N,K,splitk,launches_per_step,us_per_launch,ms_per_step,TFLOPs,GBs,roofline_us,epilogue\n')
            for v, sh, cnt, fl, by, dt, _ in sorted(out, key=lambda r: -r[2] * r[5]):
                roof = max(fl / PEAK_FOR[v[0]], by / PEAK_HBM) * 1e6
                f.write(f'{v[0]},"{v[1]}",{sh[0]},{sh[1]},{sh[2]},{sh[3]},{cnt},{dt * 1e6:.1f},'
                        f'{dt * cnt * 1e3:.3f},{fl / dt / 1e12:.1f},{by / dt / 1e9:.0f},{roof:.1f},{sh[4]}\n')
    return agg, len(rec) + len(grp) // GROUP_STEPS


PMC_FILE = 'r03_pmc_traffic.json'


def pmc_traffic(kernel):
    """-> (HBM bytes per launch of `kernel`, provenance) from the rocprofv3 PMC passes committed under profiles/: FETCH_SIZE x 2
    (the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, two SEPARATE --pmc runs of this same bench command,
    summarised by tools/pmc_summary.py (which refuses a missing pass).  Counters cannot be read from inside this process, so
    the figure is the one on file -- its commit and date travel with it; (None, None) when the file has no complete
    (read AND write) measurement of this kernel."""
    path = os.path.join(ROOT, 'profiles', PMC_FILE)
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None, None
    ent = doc.get('kernels', {}).get(kernel)
    if not ent or not ent.get('write_kib_per_launch') or not ent.get('fetch_kib_raw_per_launch'):
        return None, None
    return ent['hbm_bytes_per_launch'], {'file': 'profiles/' + PMC_FILE, 'commit': doc.get('commit'), 'collected': doc.get('collected'),
                                         'fetch_bytes': int(2048 * ent['fetch_kib_raw_per_launch']),
                                         'write_bytes': int(1024 * ent['write_kib_per_launch'])}


def attn_profile(engine, batch, reps=4):
    """Per-launch GPU time of every distinct window-attention launch of one training step (the W-MSA kernels of the
    north star), measured like gemm_profile: signatures recorded during one eager step, then `reps` launches -- each on
    its OWN operands -- inside a captured HIP graph, HIP events on the capture stream.

    Counted FLOPs: 4*64*64*D per (window, head, key tile) forward -- QK^T and AV only; softmax, the
    relative-position bias and the learned-frequency-selection filter are NOT counted -- and twice that backward.
    Algorithmic bytes: q, k, v read once and the output + log-sum-exp written once per item forward; q, k, v, dO (and O for
    the two-key-tile encoder form) read, dq, dk, dv written backward."""
    from fwair import functional as Fn
    from fwair import ops
    from fwair.lib import call as real_call, dt as dtc
    rec = []

    def spy(name, *a):
        if name in ('fw_attn_fwd', 'fw_attn_bwd'):
            if name == 'fw_attn_fwd':
                (dty, D, nkt, lfs), ld, (B, H, W, heads, L, mode, shift) = a[:4], a[7], a[14:21]
            else:
                (dty, D, nkt, lfs), ld, (B, H, W, heads, L, mode, shift) = a[:4], a[7], a[24:31]
            rec.append((name, dty, D, nkt, lfs, int(ld), B, H, W, heads, L, mode, shift))
        return real_call(name, *a)

    Fn.call = spy
    try:
        engine.step_eager(*batch)
        torch.cuda.synchronize()
    finally:
        Fn.call = real_call
    counts = {}
    for sig in rec:
        counts[sig] = counts.get(sig, 0) + 1
    side = torch.cuda.Stream()
    rows_out = []
    for sig, cnt in counts.items():
        name, dty, D, nkt, lfs, ld, B, H, W, heads, L, mode, shift = sig
        tdt = torch.bfloat16 if dty == 1 else torch.float32
        sz = 2 if dty == 1 else 4
        C = heads * D
        Cp = (C + 7) // 8 * 8
        rows = L * B * H * W
        items = B * (H // 8) * (W // 8) * L * heads
        ntab = L * L if L > 1 else 1

        def operands():
            qkv = (torch.randn(rows, ld, device='cuda') * 0.5).to(tdt)
            tables = torch.randn(ntab, 225, heads, device='cuda') * 0.2
            coef = None
            if lfs:
                coef = torch.tensor([1.1, -0.1 / 64, 0.2], device='cuda').repeat(B, heads, 1).contiguous()
            tab = ops._lfs.device_table(tdt, qkv.device) if lfs == 2 else None
            out = Fn.act_empty(rows, C, tdt, qkv.device)
            lse = torch.empty((items, 64), dtype=torch.float32, device='cuda')
            dout = Fn.act_empty(rows, C, tdt, qkv.device)
            dout.copy_((torch.randn(rows, C, device='cuda') * 0.1).to(tdt))
            dqkv = Fn.act_empty(rows, qkv.shape[1], tdt, qkv.device)
            d2 = Fn.act_empty(rows, qkv.shape[1], tdt, qkv.device) if nkt == 2 else None
            dtab = torch.zeros_like(tables)
            dcoef = torch.zeros_like(coef) if coef is not None else None
            return qkv, tables, coef, tab, out, lse, dout, dqkv, d2, dtab, dcoef

        def fwd(o):
            qkv, tables, coef, tab, out, lse = o[:6]
            real_call('fw_attn_fwd', dty, D, nkt, lfs, qkv, qkv[:, Cp:], qkv[:, Cp + C:], qkv.stride(0), out, out.stride(0), lse, tables,
                      coef, tab, B, H, W, heads, L, mode, shift, float(D) ** -0.5)

        def bwd(o):
            qkv, tables, coef, tab, out, lse, dout, dqkv, d2, dtab, dcoef = o
            real_call('fw_attn_bwd', dty, D, nkt, lfs, qkv, qkv[:, Cp:], qkv[:, Cp + C:], qkv.stride(0), out, out.stride(0), dout,
                      dout.stride(0), lse, tables, coef, tab, dqkv, dqkv[:, Cp:], dqkv[:, Cp + C:],
                      d2[:, Cp:] if d2 is not None else None, d2[:, Cp + C:] if d2 is not None else None, dqkv.stride(0), dtab, dcoef,
                      B, H, W, heads, L, mode, shift, float(D) ** -0.5, Cp - C)

        sets = [operands() for _ in range(reps)]
        run = fwd if name == 'fw_attn_fwd' else bwd
        with torch.cuda.stream(side):
            for o in sets:
                fwd(o)                                        # the backward needs a real forward (O, LSE) behind it
            run(sets[0])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for o in sets:
                    run(o)
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            g.replay()
            e1.record(side)
            torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        tile = items * 64 * D * sz                           # bytes of one [items][64][D] operand
        if name == 'fw_attn_fwd':
            fl = 4.0 * 64 * 64 * D * items * nkt
            by = tile * (1 + 2 * nkt) + tile + items * 64 * 4
        else:
            fl = 8.0 * 64 * 64 * D * items * nkt
            by = tile * (2 + 2 * nkt) + (tile if nkt > 1 else 0) + items * 64 * 4 + tile * (1 + 2 * nkt)
        kind = ('decoder' if L == 1 and D == 56 else 'encoder') + ('+lfs' if lfs == 2 else '+affine' if lfs == 1 else '')
        rows_out.append((kind, 'fwd' if name == 'fw_attn_fwd' else 'bwd', sig, cnt, fl, by, t))
        del g, sets
    return rows_out


def wmsa_block(rows, peak):
    """Aggregate attn_profile rows -> {kernel family: {fwd, bwd: time, counted TFLOP/s, MFMA fraction, GB/s, HBM fraction}}."""
    out = {}
    for kind, way, sig, cnt, fl, by, t in rows:
        d = out.setdefault(kind, {}).setdefault(way, [0.0, 0.0, 0.0, 0])
        d[0] += fl * cnt; d[1] += by * cnt; d[2] += t * cnt; d[3] += cnt
    res = {}
    for kind, ways in out.items():
        res[kind] = {}
        for way, (fl, by, t, cnt) in ways.items():
            res[kind][way] = {'launches_per_step': cnt, 'ms_per_step': round(t * 1e3, 3), 'counted_tflops': round(fl / t / 1e12, 2),
                              'mfma_frac': round(fl / t / peak, 5), 'algorithmic_gbs': round(by / t / 1e9, 1),
                              'hbm_frac': round(by / t / PEAK_HBM, 4)}
    return res


def allinone_batch(B, size, seed):
    """BASELINE configs[2] / SURVEY 8(d) config 3: denoise sigma 15 / 25 / 50, derain, dehaze cycled over the batch (the rain
    and haze pairs are the synthetic stand-ins of fwair/augment.py; the reference only reads them from disk)."""
    from fwair import augment as A
    clean, _, _ = synth_batch(B, size, 25, seed, 'cpu')
    tasks = ['denoising_15', 'denoising_25', 'denoising_50', 'deraining', 'dehazing']
    g = torch.Generator().manual_seed(seed)
    cu8 = (clean * 255).round().to(torch.uint8)
    q = torch.stack([A.degrade(cu8[i], tasks[i % 5], g).float() / 255 for i in range(B)])
    k = torch.stack([A.degrade(cu8[i], tasks[i % 5], g).float() / 255 for i in range(B)])
    return clean, q, k


def cpu_baseline(threads):
    """The CPU oracle (oracle/airnet_oracle.py, the fp32 restatement pinned against the reference) timed on the host cores as
    SURVEY.md 8(d) / BASELINE.md section 3 prescribe: the all-in-one batch B = 5 (denoise 15 / 25 / 50, derain, dehaze), 128x128,
    phase-2 steps (forward + backward + Adam lr 2e-4), 1 warm-up + 3 timed steps.  A progress line after every step."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import airnet_oracle as O
    from helpers import schema
    torch.set_num_threads(threads)
    cpu = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            cpu = next((ln.split(':', 1)[1].strip() for ln in f if ln.startswith('model name')), cpu)
    except OSError:
        pass
    B, warm, timed = 5, 1, 3
    log(f'cpu baseline: oracle train steps, B={B}, on {threads} threads of {cpu}')
    opt = O.make_opt(batch_size=B)
    st = O.fill_state_seeded(schema('all3'))
    st['E.E.queue'] = torch.nn.functional.normalize(O.seeded_tensor('E.E.queue', (3, 256, 3 * B)) / 0.02, dim=1)
    names = [k for k in st if st[k] is not None and st[k].is_floating_point() and O.is_parameter_key(k) and not k.startswith('E.E.encoder_k.')]
    for n in names:
        st[n] = st[n].clone().requires_grad_(True)
    optim = torch.optim.Adam([st[n] for n in names], lr=2e-4)
    clean, q, k = allinone_batch(B, 128, 99)
    t0 = time.time()
    for step in range(warm + timed):
        if step == warm:
            t0 = time.time()           # the first step is warm-up (allocator, thread pools)
        t1 = time.time()
        optim.zero_grad()
        restored, logits, labels = O.airnet_forward(st, opt, q, k, True)
        loss, _, _ = O.training_loss(opt, restored, logits, labels, clean)
        loss.backward()
        optim.step()
        log(f'cpu baseline: step {step + 1}/{warm + timed} took {time.time() - t1:.1f}s (loss {float(loss):.4f})')
    dt = time.time() - t0
    return {'value': round(B * timed / dt, 4), 'unit': 'images/sec', 'cores': threads, 'kind': 'port', 'cpu_model': cpu,
            'sample': f'{timed} phase-2 train steps (fwd+bwd+Adam) of the CPU oracle on the all-in-one batch B={B} '
                      f'(denoise 15/25/50, derain, dehaze), 128x128, fp32, after {warm} warm-up step'}


def log(msg):
    print(f'[bench] {msg}', file=sys.stderr, flush=True)


def host_threads():
    """Cores this process may actually use (the GPU box gives one GPU a 16-core share of a much larger host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def launch_ranks(n):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves -- one fresh child process per GPU
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1 -- BEFORE this process has made any HIP call
    (importing torch does not initialise the device), wait for them and leave with the worst exit code.  Rank 0's stdout (the
    one JSON line) is ours; the other ranks' stdout goes to stderr."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), FW_BENCH_CHILD='1')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:                    # a dead rank leaves the others waiting in a collective: stop them
                        q.terminate()
            time.sleep(0.2)
    finally:
        for q in procs:
            q.kill()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--patch-size', type=int, default=128, help='side of the training patch (128 = the headline config; 256 = SURVEY 8f-4)')
    ap.add_argument('--encoder', default='Uformer', choices=['Uformer', 'ViT'], help='ViT = BASELINE configs[4] (with --patch-size 256)')
    ap.add_argument('--tasks', default='denoise', choices=['denoise', 'allinone'], help='allinone = BASELINE configs[2]/[3]: 5 tasks cycled over the batch')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        launch_ranks(args.gpus)                               # never returns

    from fwair import engine as E
    rank, local, world = E.init_distributed()
    assert world == max(1, args.gpus), f'--gpus {args.gpus} but WORLD_SIZE={world}'
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    from net.model import AirNet
    torch.manual_seed(1234)
    tasks = ['denoising_15', 'denoising_25', 'denoising_50', 'deraining', 'dehazing'] if args.tasks == 'allinone' else None
    opt = make_opt(args.batch, args.dtype, args.patch_size, args.encoder, tasks)
    net = AirNet(opt).to(dev).train()
    eng = E.TrainEngine(net, lr=2e-4, contrast_loss_weight=0.6, use_graph=not args.no_graph)
    if tasks is None:
        batch = synth_batch(args.batch, args.patch_size, 25, 1234 + rank, dev)
    else:                                                    # BASELINE configs[2] / [3]: denoise 15 / 25 / 50, derain, dehaze cycled over the batch
        from fwair.synthetic import synth_task_batch
        batch = synth_task_batch(args.batch, args.patch_size, tasks, 1234 + rank, dev)
    clean, xq, xk = batch
    data = (xq, xk, clean)

    log(f'model built on {dev}; warm-up / graph capture ...')
    graph_ok = not args.no_graph
    try:
        for _ in range(max(1, args.warmup)):
            out = eng.step(*data)
    except Exception as e:                                   # capture problems must not cost the measurement
        if args.no_graph:
            raise
        print(f'[bench] graph capture failed ({type(e).__name__}: {e}); falling back to eager launches', file=sys.stderr)
        graph_ok = False
        eng.use_graph = False
        for _ in range(max(1, args.warmup)):
            out = eng.step(*data)
    torch.cuda.synchronize()
    log('warm-up done; timing ...')
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = eng.step(*data)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    loss = [float(v) for v in out]
    ips = args.batch * world * args.steps / dt
    log(f'{ips:.1f} images/sec, {dt / args.steps * 1e3:.1f} ms/step')
    # what the rate would be if every step also took its batch over PCIe (pinned host -> HBM, not overlapped): reported, never `value`
    host = [t.cpu().pin_memory() for t in (xq, xk, clean)]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        for h, d in zip(host, (xq, xk, clean)):
            d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    h2d = (time.perf_counter() - t1) / 5

    res = {
        'metric': f'training images/sec @{args.patch_size}x{args.patch_size}', 'value': round(ips, 2), 'unit': 'images/sec', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': (('BASELINE configs[4]: ViT encoder (N = %d tokens, Dropout 0.1 on) + plain Uformer decoder, ' % ((args.patch_size // 16) ** 2)
                                 if args.encoder == 'ViT' else
                                 ('BASELINE configs[2]: ' if tasks else 'BASELINE configs[1]: ') + 'Uformer encoder+decoder (all_3_bands, L=3, freq MSA), ')
                                + ('all-in-one (denoise 15/25/50 + derain + dehaze), ' if tasks else 'denoise sigma=25, ')
                                + f'{args.patch_size}x{args.patch_size}, phase-2 train step (fwd+bwd+Adam, DropPath on)'), 'per_gpu_batch': args.batch,
                   'global_batch': args.batch * world, 'parallelism': f'dp{world}', 'hip_graph': graph_ok},
        'loss': {'total': loss[0], 'l1': loss[1], 'contrast': loss[2]},
        'pcie_inclusive_value': round(args.batch * world / (dt / args.steps + h2d), 2), 'h2d_ms_per_step': round(h2d * 1e3, 3),
    }
    peak = PEAK_BF16 if args.dtype == 'bf16' else PEAK_F32_MFMA
    res['step_mfma_fraction'] = round(ips / world * flop_per_image_step(args.encoder, args.patch_size) / peak, 5)
    if rank == 0 and world == 1 and not args.no_profile:
        log('timing every distinct GEMM launch of the step (HIP events around captured replays) ...')
        agg, launches = gemm_profile(eng, data)
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        tot_t = sum(v[1] for v in agg.values())
        v, (fl, tt, cnt, by, troof, over) = dom
        # the variant's launches are priced one by one against max(FLOPs / MFMA peak, bytes / HBM peak); `bound` is the
        # side that sets most of that time, `achieved` / `peak` are quoted in its unit, `frac` = roofline time / measured
        hbm = by / PEAK_HBM > fl / peak
        kname = v[1]
        traffic, prov = pmc_traffic(kname)
        ach = (by / tt / 1e9) if hbm else (fl / tt / 1e12)
        pk = (PEAK_HBM / 1e9) if hbm else (peak / 1e12)
        res['roofline'] = {'bound': 'hbm' if hbm else 'mfma', 'kernel': kname,
                           'achieved': round(ach, 2), 'peak': pk, 'unit': 'GB/s' if hbm else 'TFLOP/s',
                           'frac': round(ach / pk, 5),                        # = achieved / peak, strict algorithmic bytes (X + W + C once)
                           'traffic': traffic, 'traffic_source': prov,
                           'algorithmic_bytes_per_launch': int(by / cnt), 'splitk_overhead_bytes_per_launch': int(over / cnt),
                           'avg_launch_us': round(tt / cnt * 1e6, 2), 'tflops': round(fl / tt / 1e12, 2),
                           'mfma_frac': round(fl / tt / peak, 5), 'priced_max_hbm_mfma_frac': round(troof / tt, 5),
                           'launches_per_step': cnt, 'gemm_launches_per_step': launches,
                           'gemm_time_ms_per_step': round(tot_t * 1e3, 3),
                           'all_gemm_tflops': round(sum(x[0] for x in agg.values()) / tot_t / 1e12, 2),
                           'all_gemm_roofline_frac': round(sum(x[4] for x in agg.values()) / tot_t, 5),
                           'all_gemm_splitk_overhead_gb_per_step': round(sum(x[5] for x in agg.values()) / 1e9, 3)}
        log('timing every distinct window-attention launch of the step ...')
        arows = attn_profile(eng, data)
        res['wmsa'] = wmsa_block(arows, peak)
        res['wmsa']['note'] = ('counted FLOPs = QK^T + AV only (4*64*64*D per head-window fwd, 2x bwd; SURVEY 8d); mfma_frac = counted / time / '
                               f'{peak / 1e12:.0f} TFLOP/s; hbm_frac = algorithmic bytes / time / {PEAK_HBM / 1e12:.0f} TB/s')
        dumpa = os.environ.get('FW_ATTN_DUMP')
        if dumpa:
            with open(dumpa, 'w') as f:
                f.write('kind,pass,dtype,D,nkt,lfs,B,H,W,heads,L,mode,shift,launches_per_step,us_per_launch,counted_TFLOPs,GBs\n')
                for kind, way, sig, cnt_, fl_, by_, t_ in sorted(arows, key=lambda r: -r[3] * r[6]):
                    f.write(f'{kind},{way},{sig[1]},{sig[2]},{sig[3]},{sig[4]},{sig[6]},{sig[7]},{sig[8]},{sig[9]},{sig[10]},{sig[11]},{sig[12]},'
                            f'{cnt_},{t_ * 1e6:.1f},{fl_ / t_ / 1e12:.1f},{by_ / t_ / 1e9:.0f}\n')
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(host_threads())
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
