#!/usr/bin/env python3
"""Time the fused LeFF forward kernel (fw_leff_fwd) against the unfused chain on the high-resolution stages of the B = 16 step.
FW_LEFF_NOSTORE=1 (probe only) skips the four twin stores to show what they cost."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'frequency-wised_all-in-one_image_restoration_model_amd'))
from fwair import functional as Fn  # noqa: E402
from fwair import modules as Mo  # noqa: E402

dev = 'cuda'
Fn.config.compute_dtype = torch.bfloat16


def timeit(fn, reps=5):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for C, B, H in ((112, 16, 128), (56, 16, 128), (112, 16, 64), (28, 48, 128), (56, 48, 64), (112, 48, 32)):
    leff = Mo.LeFF(C, 4 * C).to(dev)
    rows = B * H * H
    xn = Fn.act_empty(rows, C, torch.bfloat16, dev)
    xn.copy_((torch.randn(rows, C, device=dev)).to(torch.bfloat16))
    res = torch.randn(rows, C, device=dev)
    t = {}
    with torch.no_grad():
        for name, thr in (('fused', 0), ('unfused', 1 << 40)):
            Mo._LEFF_FUSED_MIN_ROWS = thr
            t[name] = timeit(lambda: leff.run(xn, res, None, B))
    hb = rows * 4 * C * 2
    print(f'C={C:3d} tokens={rows:7d}  fused {t["fused"]:7.1f} us ({4 * hb / t["fused"] / 1e6:5.2f} TB/s of twin stores)   unfused chain {t["unfused"]:7.1f} us')
