#!/usr/bin/env python3
"""Launch a few representative kernels a handful of times (for rocprofv3 --pmc runs: where do the waves spend their cycles?)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev, dt = 'cuda', torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
M, N, K = 262144, 448, 112
x = torch.randn(M, K, device=dev).to(dt)
w = (torch.randn(N, K, device=dev) * 0.1).to(dt)
y = torch.empty(M, N, device=dev, dtype=dt)
g = torch.empty(M, N, device=dev, dtype=dt)
bias = torch.randn(N, device=dev)
for _ in range(5):
    if which in ('all', 'stream_c2'):
        ops.gemm(x, w, M, N, K, out=y, bias=bias, out_gelu=g)          # stream kernel, LeFF linear1 with its GELU twin
    if which in ('all', 'stream_plain'):
        ops.gemm(x, w, M, N, K, out=y)
M2, N2, K2 = 1792, 448, 16384
dy = torch.randn(K2, M2, device=dev).to(dt)
xa = torch.randn(K2, N2, device=dev).to(dt)
dw = torch.zeros(M2, N2, device=dev)
db = torch.zeros(M2, device=dev)
for _ in range(5):
    if which in ('all', 'tn'):
        ops.wgrad(dy, xa, M2, N2, K2, dw, db)
xb = torch.randn(16384, 448, device=dev).to(dt)
wb = (torch.randn(1792, 448, device=dev) * 0.1).to(dt)
yb = torch.empty(16384, 1792, device=dev, dtype=dt)
for _ in range(5):
    if which in ('all', 'nt'):
        ops.gemm(xb, wb, 16384, 1792, 448, out=yb)
torch.cuda.synchronize()
