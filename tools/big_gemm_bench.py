#!/usr/bin/env python3
"""A/B of the compute-bound GEMM shapes of the B = 16 step: FW_GEMM_BIG=0 (128 x 128 rings) vs 1 / 2 / 3 (256 x 256 tiles, 8 waves).
Each shape is timed as 8 launches on 8 separate operand sets inside a captured graph (cold L2 / MALL like in the step)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev = 'cuda'
bf = torch.bfloat16
SHAPES = [  # (M, N, K, kind)   kind: fc1 = bias + GELU twin, plain = bias, dgelu = dX with GELU' epilogue (W [K][N]), dplain
    (16384, 1792, 448, 'fc1'), (4096, 3584, 896, 'fc1'), (16384, 1344, 448, 'plain'), (4096, 2688, 896, 'plain'), (1024, 65536, 448, 'plain'),
    (16384, 1792, 448, 'dgelu'), (4096, 3584, 896, 'dgelu'), (16384, 448, 1792, 'dplain'), (4096, 896, 3584, 'dplain'),
    (16384, 448, 1344, 'dplain'), (1024, 448, 65536, 'dplain'),
]


def run(M, N, K, kind, reps=8):
    sets = []
    for _ in range(reps):
        if kind in ('fc1', 'plain'):
            x = (torch.randn(M, K, device=dev) * 0.5).to(bf)
            w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
            b = torch.randn(N, device=dev)
            y = torch.empty(M, N, device=dev, dtype=bf)
            g = torch.empty(M, N, device=dev, dtype=bf) if kind == 'fc1' else None
            sets.append(lambda x=x, w=w, b=b, y=y, g=g: ops.gemm(x, w, M, N, K, out=y, bias=b, out_gelu=g))
        else:
            dy = (torch.randn(M, K, device=dev) * 0.5).to(bf)               # reduction = K here
            w = (torch.randn(K, N, device=dev) * 0.05).to(bf)
            out = torch.empty(M, N, device=dev, dtype=bf)
            aux = (torch.randn(M, N, device=dev)).to(bf) if kind == 'dgelu' else None
            sets.append(lambda dy=dy, w=w, out=out, aux=aux: ops.gemm(dy, w, M, N, K, w_trans=True, out=out, act=2 if aux is not None else 0, aux=aux))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for f in sets:
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for f in sets:
                f()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(3):
            g.replay()
        e1.record(side)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (3 * reps)


if __name__ == '__main__':
    print('FW_GEMM_BIG =', os.environ.get('FW_GEMM_BIG', '(default 1)'))
    tot = 0.0
    for M, N, K, kind in SHAPES:
        t = run(M, N, K, kind)
        tot += t
        print(f'{kind:7s} M={M:6d} N={N:6d} K={K:6d}  {t * 1e6:8.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s')
    print(f'sum {tot * 1e6:.1f} us')
