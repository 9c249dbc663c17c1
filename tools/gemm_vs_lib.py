#!/usr/bin/env python3
"""fw_gemm against the vendor library (torch.mm -> hipBLASLt / rocBLAS) on the mid-size shapes of the step: what a tuned
library reaches on MI355X for the same bf16 problem is the practical ceiling the hand-written tile kernel is compared with."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev = 'cuda'
dt = torch.bfloat16


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


shapes = [(16384, 1792, 448), (16384, 448, 1792), (16384, 448, 448), (16384, 896, 448), (4096, 3584, 896), (4096, 896, 3584), (4096, 896, 896),
          (4096, 1792, 896), (1024, 65536, 448), (65536, 224, 224), (65536, 896, 224), (1024, 896, 896), (1024, 3584, 896), (1024, 7168, 896)]
print(f'{"form":4s} {"M":>7s} {"N":>6s} {"K":>6s} {"fw us":>9s} {"fw TF/s":>8s} {"lib us":>9s} {"lib TF/s":>8s}')
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.1).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    wt = w.t().contiguous()
    fl = 2.0 * M * N * K
    t1 = timeit(lambda: ops.gemm(x, w, M, N, K, out=y))
    t2 = timeit(lambda: torch.mm(x, w.t(), out=y))
    print(f'NT   {M:7d} {N:6d} {K:6d} {t1 * 1e6:9.1f} {fl / t1 / 1e12:8.1f} {t2 * 1e6:9.1f} {fl / t2 / 1e12:8.1f}')
    g = torch.randn(M, N, device=dev).to(dt)
    dx = torch.empty(M, K, device=dev, dtype=dt)
    t1 = timeit(lambda: ops.gemm(g, w, M, K, N, w_trans=True, out=dx))
    t2 = timeit(lambda: torch.mm(g, w, out=dx))
    print(f'NN   {M:7d} {K:6d} {N:6d} {t1 * 1e6:9.1f} {fl / t1 / 1e12:8.1f} {t2 * 1e6:9.1f} {fl / t2 / 1e12:8.1f}')
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    dwb = torch.empty(N, K, device=dev, dtype=dt)
    t1 = timeit(lambda: ops.wgrad(g, x, N, K, M, dw, db))
    t2 = timeit(lambda: torch.mm(g.t(), x, out=dwb))
    print(f'TN   {N:7d} {K:6d} {M:6d} {t1 * 1e6:9.1f} {fl / t1 / 1e12:8.1f} {t2 * 1e6:9.1f} {fl / t2 / 1e12:8.1f}')
