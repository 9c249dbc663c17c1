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


def synth_batch(B, size, sigma, seed, device):
    """low-frequency cosines + rectangles, uint8-quantised; noise as utils/dataset_utils.py:126."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    clean = np.zeros((B, 3, size, size), np.float32)
    for b in range(B):
        for c in range(3):
            img = np.zeros((size, size), np.float32)
            for _ in range(8):
                fx, fy, ph, a = rs.uniform(0, 4), rs.uniform(0, 4), rs.uniform(0, 6.28), rs.uniform(0.2, 1)
                img += a * np.cos(6.2832 * (fx * xx + fy * yy) + ph)
            for _ in range(4):
                x0, y0 = rs.randint(0, size - 8, 2)
                w, h = rs.randint(8, size // 2, 2)
                img[y0:y0 + h, x0:x0 + w] += rs.uniform(-1, 1)
            img = (img - img.min()) / max(img.max() - img.min(), 1e-6)
            clean[b, c] = np.round(img * 255) / 255
    def noisy():
        return np.clip(clean * 255 + sigma * rs.randn(*clean.shape), 0, 255).astype(np.uint8).astype(np.float32) / 255
    t = lambda a: torch.from_numpy(a).to(device)
    return t(clean), t(noisy()), t(noisy())


def make_opt(batch, dtype, patch=128):
    import types
    return types.SimpleNamespace(L=3, encoder_dim=256, encoder_embed_dim=28, embed_dim=56, batch_size=batch, patch_size=patch,
                                 degradation_embedding_method=['all_3_bands'], encoder_msa_type='freq', contrast_loss_weight=0.6,
                                 encoder_type='Uformer', decoder_type='Uformer', debug_mode=False, frequency_decompose_type='none',
                                 learnable_modulator=False, compute_dtype=dtype, de_type=['denoising_25'] * batch)


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

    ops.gemm = hook
    try:
        engine.step_eager(*batch)
        torch.cuda.synchronize()
    finally:
        ops.gemm = orig
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
            code = L.fw_gemm_last_variant()                   # the kernel fw_gemm really chose (names as in the rocprofv3 stats)
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
        by = (M * K + N * K) * sz + M * N * kw['out'].element_size() * sk
        if kw.get('accumulate'):
            by += M * N * kw['out'].element_size()
        for key in ('residual', 'aux', 'out_gelu'):
            if kw.get(key) is not None:
                by += M * N * kw[key].element_size()
        fam, bn, xt, wt = code // 100000, code // 100 % 1000, bool(code // 10 % 10), bool(code % 10)
        variant = ('bf16' if x.dtype == torch.bfloat16 else 'f32', ('stream', 'tr')[2 - fam] if fam else bn, xt, wt)
        out.append((variant, (M, N, K, sk), cnt, 2.0 * M * N * K, float(by), dt))
        del g, sets, kw, x, w
    agg = {}
    for variant, shape, cnt, fl, by, dt in out:
        d = agg.setdefault(variant, [0.0, 0.0, 0, 0.0, 0.0])
        d[0] += fl * cnt; d[1] += dt * cnt; d[2] += cnt; d[3] += by * cnt
        d[4] += cnt * max(fl / PEAK_FOR[variant[0]], by / PEAK_HBM)          # time the launches would take at their roofline
    dump = os.environ.get('FW_GEMM_DUMP')
    if dump:
        with open(dump, 'w') as f:
            f.write('dtype,kernel,xT,wT,M,N,K,splitk,launches_per_step,us_per_launch,ms_per_step,TFLOPs,GBs,roofline_us\n')
            for v, sh, cnt, fl, by, dt in sorted(out, key=lambda r: -r[2] * r[5]):
                roof = max(fl / PEAK_FOR[v[0]], by / PEAK_HBM) * 1e6
                f.write(f'{v[0]},{v[1]},{int(v[2])},{int(v[3])},{sh[0]},{sh[1]},{sh[2]},{sh[3]},{cnt},{dt * 1e6:.1f},'
                        f'{dt * cnt * 1e3:.3f},{fl / dt / 1e12:.1f},{by / dt / 1e9:.0f},{roof:.1f}\n')
    return agg, len(rec)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x 2 -- the gfx950
    correction of MI355X_MICROARCH.md -- plus WRITE_SIZE, two separate --pmc runs of this same bench command; summarised by
    tools/pmc_summary.py).  None when no measurement for this kernel is on file."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get('hbm_bytes_per_launch')
    except (OSError, ValueError):
        return None


def cpu_baseline(threads):
    """The CPU oracle (oracle/airnet_oracle.py, the fp32 restatement pinned against the reference) timed on the host
    cores: phase-2 steps (forward + backward + Adam) at B = 2, 128x128 -- a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import airnet_oracle as O
    from helpers import schema
    torch.set_num_threads(threads)
    log(f'cpu baseline: oracle train steps on {threads} threads')
    B = 2
    opt = O.make_opt(batch_size=B)
    st = O.fill_state_seeded(schema('all3'))
    names = [k for k in st if st[k] is not None and st[k].is_floating_point() and O.is_parameter_key(k) and not k.startswith('E.E.encoder_k.')]
    for n in names:
        st[n] = st[n].clone().requires_grad_(True)
    optim = torch.optim.Adam([st[n] for n in names], lr=2e-4)
    clean, q, k = (t.cpu() for t in synth_batch(B, 128, 25, 99, 'cpu'))
    t0 = time.time()
    steps = 0
    while steps < 2:
        optim.zero_grad()
        restored, logits, labels = O.airnet_forward(st, opt, q, k, True)
        loss, _, _ = O.training_loss(opt, restored, logits, labels, clean)
        loss.backward()
        optim.step()
        steps += 1
        log(f'cpu baseline: step {steps} done at {time.time() - t0:.1f}s')
        if steps == 1:
            t0 = time.time()           # first step = warm-up (allocator, thread pools)
    dt = time.time() - t0
    return {'value': round(B * (steps - 1) / dt, 4), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'{steps - 1} phase-2 train step(s) (fwd+bwd+Adam) of the CPU oracle, B={B}, 128x128, fp32, after 1 warm-up step'}


def log(msg):
    print(f'[bench] {msg}', file=sys.stderr, flush=True)


def host_threads():
    """Cores this process may actually use (the GPU box gives one GPU a 16-core share of a much larger host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--patch-size', type=int, default=128, help='side of the training patch (128 = the headline config; 256 = SURVEY 8f-4)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    args = ap.parse_args()

    from fwair import engine as E
    rank, local, world = E.init_distributed()
    assert world == max(1, args.gpus) or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    from net.model import AirNet
    torch.manual_seed(1234)
    opt = make_opt(args.batch, args.dtype, args.patch_size)
    net = AirNet(opt).to(dev).train()
    eng = E.TrainEngine(net, lr=2e-4, contrast_loss_weight=0.6, use_graph=not args.no_graph)
    batch = synth_batch(args.batch, args.patch_size, 25, 1234 + rank, dev)
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
        'config': {'workload': 'BASELINE configs[1]: Uformer encoder+decoder (all_3_bands, L=3, freq MSA), denoise sigma=25, '
                               f'{args.patch_size}x{args.patch_size}, phase-2 train step (fwd+bwd+Adam, DropPath on)', 'per_gpu_batch': args.batch,
                   'global_batch': args.batch * world, 'parallelism': f'dp{world}', 'hip_graph': graph_ok},
        'loss': {'total': loss[0], 'l1': loss[1], 'contrast': loss[2]},
        'pcie_inclusive_value': round(args.batch * world / (dt / args.steps + h2d), 2), 'h2d_ms_per_step': round(h2d * 1e3, 3),
    }
    peak = PEAK_BF16 if args.dtype == 'bf16' else PEAK_F32_MFMA
    res['step_mfma_fraction'] = round(ips / world * FLOP_PER_IMAGE_STEP * (args.patch_size / 128) ** 2 / peak, 5)   # FLOPs scale with pixels
    if rank == 0 and world == 1 and not args.no_profile:
        log('timing every distinct GEMM launch of the step (HIP events around captured replays) ...')
        agg, launches = gemm_profile(eng, data)
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        tot_t = sum(v[1] for v in agg.values())
        v, (fl, tt, cnt, by, troof) = dom
        # the variant's launches are priced one by one against max(FLOPs / MFMA peak, bytes / HBM peak); `bound` is the
        # side that sets most of that time, `achieved` / `peak` are quoted in its unit, `frac` = roofline time / measured
        hbm = by / PEAK_HBM > fl / peak
        kname = (f'gemm_stream_kernel<{v[0]},wT={int(v[3])}>' if v[1] == 'stream' else f'gemm_tr_kernel<xT={int(v[2])}>' if v[1] == 'tr'
                 else f'gemm_kernel<{v[0]},BN={v[1]},xT={int(v[2])},wT={int(v[3])}>')
        res['roofline'] = {'bound': 'hbm' if hbm else 'mfma', 'kernel': kname,
                           'achieved': round((by / tt / 1e9) if hbm else (fl / tt / 1e12), 2),
                           'peak': (PEAK_HBM / 1e9) if hbm else (peak / 1e12), 'unit': 'GB/s' if hbm else 'TFLOP/s',
                           'frac': round(troof / tt, 5), 'traffic': pmc_traffic(kname), 'avg_launch_us': round(tt / cnt * 1e6, 2),
                           'tflops': round(fl / tt / 1e12, 2), 'algorithmic_gbs': round(by / tt / 1e9, 1),
                           'launches_per_step': cnt, 'gemm_launches_per_step': launches,
                           'gemm_time_ms_per_step': round(tot_t * 1e3, 3),
                           'all_gemm_tflops': round(sum(x[0] for x in agg.values()) / tot_t / 1e12, 2),
                           'all_gemm_roofline_frac': round(sum(x[4] for x in agg.values()) / tot_t, 5)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(host_threads())
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
