#!/usr/bin/env python3
"""Tall-skinny GEMM shapes of the C <= 112 stages (gemm_stream_kernel): time per launch on cold operands (8 sets in a captured graph),
GB/s of algorithmic bytes.  FW_GEMM_BIG_DBG=1 skips the epilogue's stores (what is left is loads + MFMA)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev, bf = 'cuda', torch.bfloat16
SHAPES = [(786432, 88, 28, 'plain'), (786432, 112, 28, 'plain'), (786432, 28, 112, 'res'), (262144, 168, 56, 'plain'), (262144, 224, 56, 'plain'),
          (262144, 56, 224, 'res'), (262144, 448, 112, 'plain'), (262144, 336, 112, 'plain'), (196608, 224, 56, 'plain'),
          (786432, 28, 88, 'dplain'), (786432, 28, 112, 'dgelu_in'), (262144, 112, 448, 'dgelu'), (262144, 56, 224, 'dgelu')]


def ld8(n):
    return (n + 7) // 8 * 8


def run(M, N, K, kind, reps=8):
    sets = []
    for _ in range(reps):
        if kind in ('plain', 'res'):
            x = (torch.randn(M, ld8(K), device=dev) * 0.5).to(bf)[:, :K]
            w = (torch.randn(N, ld8(K), device=dev) * 0.05).to(bf)[:, :K]
            b = torch.randn(N, device=dev)
            if kind == 'res':
                res = torch.randn(M, N, device=dev)
                y = torch.empty(M, N, device=dev)
                sets.append(lambda x=x, w=w, b=b, y=y, res=res: ops.gemm(x, w, M, N, K, out=y, bias=b, residual=res))
            else:
                y = torch.empty(M, ld8(N), device=dev, dtype=bf)[:, :N]
                sets.append(lambda x=x, w=w, b=b, y=y: ops.gemm(x, w, M, N, K, out=y, bias=b))
        else:
            dy = (torch.randn(M, ld8(K), device=dev) * 0.5).to(bf)[:, :K]
            w = (torch.randn(K, ld8(N), device=dev) * 0.05).to(bf)[:, :N]
            out = torch.empty(M, ld8(N), device=dev, dtype=bf)[:, :N]
            aux = torch.randn(M, ld8(N), device=dev).to(bf)[:, :N] if kind.startswith('dgelu') else None
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
    tot = 0.0
    for M, N, K, kind in SHAPES:
        t = run(M, N, K, kind)
        tot += t
        by = M * K * 2 + (M * N * 4 * 2 if kind == 'res' else M * N * 2) + (M * N * 2 if kind.startswith('dgelu') else 0)
        print(f'{kind:9s} M={M:7d} N={N:4d} K={K:4d}  {t * 1e6:8.1f} us  {by / t / 1e12:5.2f} TB/s')
    print(f'sum {tot * 1e6:.1f} us')
