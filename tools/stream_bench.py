#!/usr/bin/env python3
"""What plain streaming reaches on this GPU (copy, scale, 1R:2W) for footprints above the 256 MB infinity cache."""
import torch
dev = 'cuda'


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


for mb in (64, 235, 470, 940):
    n = mb * 1024 * 1024 // 2
    a = torch.randn(n, device=dev).to(torch.bfloat16)
    b = torch.empty_like(a); c = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    t2 = timeit(lambda: torch.mul(a, 2.0, out=b))
    def two():
        torch.mul(a, 2.0, out=b); torch.mul(a, 3.0, out=c)
    t3 = timeit(two)
    t4 = timeit(lambda: a.sum())
    print(f'{mb:5d} MB: copy {2 * mb / 1024 / t / 1e3 * 1.0737:6.2f} TB/s   mul {2 * mb / 1024 / t2 / 1e3 * 1.0737:6.2f} TB/s   2x mul {4 * mb / 1024 / t3 / 1e3 * 1.0737:6.2f} TB/s   read-only sum {mb / 1024 / t4 / 1e3 * 1.0737:6.2f} TB/s')
