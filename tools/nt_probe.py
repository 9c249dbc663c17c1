#!/usr/bin/env python3
"""Time fw_gemm on the heaviest forward / input-gradient shapes of the B = 16 step (random operands, own buffers per repetition).
Kernel variants are chosen by FW_GEMM_RING / FW_GEMM_TR_RING in the environment (read once by the library)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev, dtype = 'cuda', torch.bfloat16
NT = [(16384, 1792, 448), (4096, 3584, 896), (16384, 448, 1792), (1024, 65536, 448), (4096, 896, 3584), (16384, 1344, 448), (16384, 896, 224),
      (4096, 1792, 448), (65536, 448, 224)]
NN = [(16384, 448, 1792), (4096, 896, 3584), (16384, 1792, 448), (4096, 3584, 896)]          # dx[M,N] = g[M,K] W[K,N]


def timeit(fn, sets, iters=3):
    for s in sets:
        fn(*s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        for s in sets:
            fn(*s)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (iters * len(sets)) * 1e3


print('FW_GEMM_RING', os.environ.get('FW_GEMM_RING', 'default'), 'FW_GEMM_TR_RING', os.environ.get('FW_GEMM_TR_RING', 'default'))
tot = 0.0
for M, N, K in NT:
    sets = [(torch.randn(M, K, device=dev).to(dtype), (torch.randn(N, K, device=dev) * 0.1).to(dtype), torch.empty(M, N, device=dev, dtype=dtype),
             torch.randn(N, device=dev)) for _ in range(4)]
    t = timeit(lambda x, w, y, b: ops.gemm(x, w, M, N, K, out=y, bias=b), sets)
    tot += t
    print(f'NT y=xW^T  M={M:6d} N={N:6d} K={K:5d}  {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.0f} TF/s')
for M, N, K in NN:
    sets = [(torch.randn(M, K, device=dev).to(dtype), (torch.randn(K, N, device=dev) * 0.1).to(dtype), torch.empty(M, N, device=dev, dtype=dtype))
            for _ in range(4)]
    t = timeit(lambda g, w, y: ops.gemm(g, w, M, N, K, w_trans=True, out=y), sets)
    tot += t
    print(f'NN dx=gW   M={M:6d} N={N:6d} K={K:5d}  {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.0f} TF/s')
print(f'sum {tot:.1f} us')
