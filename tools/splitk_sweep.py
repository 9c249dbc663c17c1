#!/usr/bin/env python3
"""Sweep the split-K factor of the weight-gradient GEMM (ops.wgrad) over the shapes of the B=16 step and print the best per shape."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev, dtype = 'cuda', torch.bfloat16


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


stages = [(262144, 56), (65536, 112), (16384, 224), (4096, 448), (1024, 896), (4096, 896), (16384, 448), (65536, 224), (262144, 112),
          (786432, 28), (196608, 56), (49152, 112), (12288, 224), (3072, 448)]
best = {}
for T, C in stages:
    for (N, K) in ((C, C), (2 * C, C), (4 * C, C), (C, 4 * C)):
        ldk, ldn = (K + 7) // 8 * 8, (N + 7) // 8 * 8
        x = torch.randn(T, ldk, device=dev).to(dtype)[:, :K]
        g = torch.randn(T, ldn, device=dev).to(dtype)[:, :N]
        dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        res = {}
        orig = ops.pick_splitk
        for sk in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
            if T // sk < 64 or sk * N * K * 4 > (256 << 20):
                continue
            ops.pick_splitk = lambda *a, sk=sk: sk
            res[sk] = timeit(lambda: ops.wgrad(g, x, N, K, T, dw, db))
        ops.pick_splitk = orig
        b = min(res, key=res.get)
        best[f'{N},{K},{T}'] = b
        print(f'N={N:5d} K={K:5d} T={T:7d} default sk={orig(N, K, T, dtype):4d}  best sk={b:4d} {res[b]:7.1f} us   ' +
              ' '.join(f'{k}:{v:.0f}' for k, v in res.items()))
print(json.dumps(best))
