#!/usr/bin/env python3
"""Time fw_attn_fwd / fw_attn_bwd on the window-attention launches of the B = 16 step (decoder stages with LFS, encoder intra).
FW_ATTN_V2=0 selects the v1 kernels (read once by the library):  for v in 0 1; do FW_ATTN_V2=$v python tools/attn_probe.py; done"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev, dtype = 'cuda', torch.bfloat16
# (B, H, heads, D, L, mode, lfs, shift): decoder stages 0..4 and back (one block each), encoder stage 0 intra
CASES = [(16, 128, 1, 56, 1, 0, 2, 4), (16, 128, 2, 56, 1, 0, 2, 4), (16, 64, 2, 56, 1, 0, 2, 0), (16, 64, 4, 56, 1, 0, 2, 4),
         (16, 32, 4, 56, 1, 0, 2, 4), (16, 32, 8, 56, 1, 0, 2, 0), (16, 16, 8, 56, 1, 0, 2, 4), (16, 16, 16, 56, 1, 0, 2, 0),
         (16, 8, 16, 56, 1, 0, 2, 0), (16, 128, 1, 28, 3, 0, 0, 4), (16, 128, 1, 28, 3, 1, 0, 4)]


def timeit(fn, reps=6):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print('FW_ATTN_V2', os.environ.get('FW_ATTN_V2', 'default'))
tf = tb = 0.0
for B, H, heads, D, L, mode, lfs, shift in CASES:
    C = heads * D
    rows = L * B * H * H
    ld = (3 * C + 7) // 8 * 8
    qkv = (torch.randn(rows, ld, device=dev) * 0.5).to(dtype)[:, :3 * C]
    tables = torch.randn(L * L if L > 1 else 1, 225, heads, device=dev) * 0.2
    coef = torch.tensor([1.1, -0.1 / 64, 0.2], device=dev).repeat(B, heads, 1).contiguous() if lfs else None
    out, lse = ops.attn_fwd(qkv, C, B, H, H, heads, L, mode, shift, tables, coef, lfs)
    dout = (torch.randn(rows, (C + 7) // 8 * 8, device=dev) * 0.1).to(dtype)[:, :C]
    dbias = torch.zeros_like(tables)
    dcoef = torch.zeros_like(coef) if coef is not None else None
    t1 = timeit(lambda: ops.attn_fwd(qkv, C, B, H, H, heads, L, mode, shift, tables, coef, lfs))
    t2 = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, C, B, H, H, heads, L, mode, shift, tables, dbias, coef, dcoef, lfs))
    items = B * (H // 8) ** 2 * L * heads
    nkt = 1 if mode == 0 else L - 1
    tile = items * 64 * D * 2
    fb, bb = tile * (2 + 2 * nkt) + items * 256, tile * (3 + 4 * nkt) + (tile if nkt > 1 else 0)
    tf += t1; tb += t2
    print(f'B={B} {H:3d}x{H:<3d} heads={heads:2d} D={D} L={L} mode={mode} lfs={lfs} items={items:6d}  fwd {t1:7.1f} us {fb / t1 / 1e6:6.2f} TB/s | '
          f'bwd {t2:7.1f} us {bb / t2 / 1e6:6.2f} TB/s  ({t2 * 256 / items:5.2f} us per item-slot)')
print(f'sum fwd {tf:.0f} us, bwd {tb:.0f} us')
