#!/usr/bin/env python3
"""Time the weight-gradient GEMM (ops.wgrad: dW = dY^T x, split-K slab + fold) on the heaviest shapes of the B = 16 step for a
range of split factors.  The kernel variant is chosen by FW_GEMM_TR_RING in the environment (read once by the library), so
run it once per variant:   for r in 0 1 2 3 4; do FW_GEMM_TR_RING=$r python tools/wgrad_probe.py; done"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev, dtype = 'cuda', torch.bfloat16
SHAPES = [(1344, 448, 16384), (3584, 896, 4096), (448, 1792, 16384), (224, 896, 16384), (448, 448, 16384), (448, 112, 262144),
          (896, 896, 4096), (65536, 448, 1024)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]


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


print('ring variant', os.environ.get('FW_GEMM_TR_RING', 'default'))
for N, K, T in SHAPES:
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    sets = []
    for i in range(6):                                   # own operands per repetition: nothing is found in the infinity cache
        x = torch.randn(T, K, device=dev).to(dtype)
        g = torch.randn(T, N, device=dev).to(dtype)
        sets.append((g, x, torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)))
    default = ops.pick_splitk(N, K, T, dtype)
    cands = sorted({default} | {max(1, b // tiles) for b in (128, 192, 256, 320, 384, 448, 512, 640, 768, 1024, 1536, 2048)})
    orig = ops.pick_splitk
    res = {}
    for sk in cands:
        if T // sk < 128:
            continue
        ops.pick_splitk = lambda *a, sk=sk: sk
        res[sk] = timeit(lambda g, x, dw, db: ops.wgrad(g, x, N, K, T, dw, db), sets)
    ops.pick_splitk = orig
    best = min(res, key=res.get)
    fl = 2.0 * N * K * T
    print(f'M={N:6d} N={K:5d} K={T:7d} tiles={tiles:4d} default sk={default:4d} {res.get(default, float("nan")):7.1f} us | best sk={best:4d} '
          f'{res[best]:7.1f} us ({fl / res[best] / 1e6:6.0f} TF/s) | ' + ' '.join(f'{k}({k * tiles}):{v:.0f}' for k, v in res.items()))
