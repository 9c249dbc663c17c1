#!/usr/bin/env python3
"""Time fw_dwconv_bwd (data + weight gradient of the LeFF depthwise 3x3) on the layer shapes of the B = 16 step.
Run once per setting of FW_DWCONV_FUSED_BWD (the library reads it once): 1 = one fused pass, 0 = data-gradient kernel + weight-gradient kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import ops  # noqa: E402

dev = 'cuda'
print('FW_DWCONV_FUSED_BWD =', os.environ.get('FW_DWCONV_FUSED_BWD', 'default'))
for B, H, C in ((16, 128, 112), (16, 64, 224), (16, 32, 448), (16, 16, 896), (16, 32, 896), (16, 64, 448), (16, 128, 224), (48, 128, 112)):
    rows = B * H * H
    sets = []
    for _ in range(4):                                   # rotate operand sets: no launch finds its inputs in the infinity cache
        dh2 = (torch.randn(rows, C, device=dev) * 0.5).to(torch.bfloat16)
        h1 = (torch.randn(rows, C, device=dev)).to(torch.bfloat16)
        g1 = torch.nn.functional.gelu(h1.float()).to(torch.bfloat16)
        sets.append((dh2, g1 if os.environ.get('FW_PROBE_TWIN') else None, h1))
    w = torch.randn(9, C, device=dev)
    dw, db = torch.zeros(C, 9, device=dev), torch.zeros(C, device=dev)
    for s in sets:
        ops.dwconv_bwd(*s, w, dw, db, B, H, H)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(3):
        for s in sets:
            ops.dwconv_bwd(*s, w, dw, db, B, H, H)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 12 * 1e3
    by = rows * C * 2 * 3                                # dh2 + h1 read, dh1 written
    print(f'B={B:3d} H={H:4d} C={C:4d}  {us:8.1f} us   {by / us / 1e6:5.2f} TB/s of (dh2 + h1 + dh1)')
