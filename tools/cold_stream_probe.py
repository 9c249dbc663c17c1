#!/usr/bin/env python3
"""What a SHORT streaming kernel reaches on this GPU when its operands are cold (not in the 256 MB infinity cache) -- the situation
of every kernel inside the training step.  A captured graph runs one copy / one read-modify-write kernel per buffer pair over a
ring of distinct buffers (total footprint >> 256 MB), so each launch moves `mb` MB in and `mb` MB out exactly once.
Prints the average kernel time and the rate per footprint; compare with the 5.6 TB/s of the 4.4 GB Adam launch."""
import sys

import torch

dev = 'cuda'
total_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0


def run(mb, op):
    n = mb * 1024 * 1024 // 2                                  # bf16 elements
    pairs = max(4, int(total_gb * 1024 / (2 * mb)))
    src = [torch.randn(n, device=dev).to(torch.bfloat16) for _ in range(pairs)]
    dst = [torch.empty_like(s) for s in src]
    fn = (lambda s, d: d.copy_(s)) if op == 'copy' else (lambda s, d: torch.mul(s, 1.5, out=d))
    for s, d in zip(src, dst):
        fn(s, d)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for s, d in zip(src, dst):
                fn(s, d)
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); g.replay(); b.record(st)
        torch.cuda.synchronize()
    t = a.elapsed_time(b) * 1e-3 / pairs
    return t, 2 * mb * 1024 * 1024 / t / 1e12


for mb in (8, 16, 32, 64, 128, 256, 1024):
    t1, r1 = run(mb, 'copy')
    t2, r2 = run(mb, 'mul')
    print(f'{mb:5d} MB in + {mb:5d} MB out per launch:  copy {t1 * 1e6:8.1f} us {r1:5.2f} TB/s    x*1.5 {t2 * 1e6:8.1f} us {r2:5.2f} TB/s')
