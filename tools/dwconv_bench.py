#!/usr/bin/env python3
"""Graph-timed LeFF depthwise-conv kernels (forward, data gradient + weight gradient) on the shapes of the B=16 step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev = 'cuda'
dt = torch.bfloat16


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        g.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


# (B, H, C hidden): decoder stages and the stacked-band encoder stages
shapes = [(16, 128, 448), (16, 128, 224), (16, 64, 896), (16, 64, 448), (16, 32, 1792), (16, 32, 896), (16, 16, 3584), (16, 16, 1792),
          (16, 8, 3584), (48, 128, 112), (48, 64, 224), (48, 32, 448), (48, 16, 896)]
print(f'{"B":>3s} {"H":>4s} {"C":>5s} {"fwd us":>9s} {"GB/s":>6s} {"bwd us":>9s} {"GB/s":>6s}')
for B, H, C in shapes:
    rows = B * H * H
    g1 = torch.randn(rows, C, device=dev).to(dt)
    h1 = torch.randn(rows, C, device=dev).to(dt)
    dh2 = torch.randn(rows, C, device=dev).to(dt)
    w = torch.randn(9, C, device=dev)
    b = torch.randn(C, device=dev)
    dw, db = torch.zeros(C, 9, device=dev), torch.zeros(C, device=dev)
    t1 = timeit(lambda: ops.dwconv_fwd(h1, w, b, B, H, H, in_gelu=True))
    t2 = timeit(lambda: ops.dwconv_bwd(dh2, None, h1, w, dw, db, B, H, H))
    by1 = rows * C * 2 * 3
    by2 = rows * C * 2 * 5            # data grad: dh2, h1 -> dh1; weight grad: dh2, g1
    print(f'{B:3d} {H:4d} {C:5d} {t1 * 1e6:9.1f} {by1 / t1 / 1e9:6.0f} {t2 * 1e6:9.1f} {by2 / t2 / 1e9:6.0f}')
