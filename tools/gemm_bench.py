#!/usr/bin/env python3
"""Micro-benchmark of fw_gemm on the shapes of the B=16 training step (decoder + encoder stages, heads).
Prints time, TFLOP/s and algorithmic GB/s per (variant, shape) -- used to decide where the GEMM needs work."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev = 'cuda'
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
sz = 2 if dtype == torch.bfloat16 else 4


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def report(tag, M, N, K, t, bytes_):
    print(f'{tag:34s} M={M:7d} N={N:6d} K={K:7d}  {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:8.1f} TF/s  {bytes_ / t / 1e9:8.0f} GB/s')


stages = [(262144, 56), (65536, 112), (16384, 224), (4096, 448), (1024, 896), (4096, 896), (16384, 448), (65536, 224), (262144, 112),
          (786432, 28), (196608, 56), (49152, 112), (12288, 224), (3072, 448)]
tot = {'NT': 0.0, 'NN': 0.0, 'TN': 0.0}
skinny = len(sys.argv) > 2 and sys.argv[2] == 'skinny'          # only the shapes of gemm_stream_kernel, no weight grads
if skinny:
    stages = [s for s in stages if s[0] >= 32768]
for T, C in stages:
    for (N, K) in ((C, C), (2 * C, C), (4 * C, C), (C, 4 * C)):
        ldk = (K + 7) // 8 * 8
        ldn = (N + 7) // 8 * 8
        x = torch.randn(T, ldk, device=dev).to(dtype)[:, :K]
        w = (torch.randn(N, ldk, device=dev) * 0.1).to(dtype)[:, :K]
        y = torch.empty(T, ldn, device=dev, dtype=dtype)[:, :N]
        t = timeit(lambda: ops.gemm(x, w, T, N, K, out=y))
        report(f'NT  y=xW^T C={C}', T, N, K, t, (T * K + N * K + T * N) * sz)
        tot['NT'] += t
        g = torch.randn(T, ldn, device=dev).to(dtype)[:, :N]
        dx = torch.empty(T, ldk, device=dev, dtype=dtype)[:, :K]
        t = timeit(lambda: ops.gemm(g, w, T, K, N, w_trans=True, out=dx))
        report(f'NN  dx=gW C={C}', T, K, N, t, (T * K + N * K + T * N) * sz)
        tot['NN'] += t
        if skinny:
            continue
        dw = torch.zeros(N, K, device=dev)
        sk = ops.pick_splitk(N, K, T, dtype)
        db = torch.zeros(N, device=dev)
        t = timeit(lambda: ops.wgrad(g, x, N, K, T, dw, db))
        report(f'TN  dW=g^Tx C={C} splitk={sk}', N, K, T, t, (T * K + T * N) * sz + N * K * 4)
        tot['TN'] += t
print({k: round(v * 1e3, 2) for k, v in tot.items()}, 'ms (one instance of each shape)')
