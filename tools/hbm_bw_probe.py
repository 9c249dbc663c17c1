import torch, time
dev='cuda'
def t(fn, n=20):
    for _ in range(3): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e-3
for mb in (64, 256, 1024, 4096):
    n = mb*1024*1024//4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    tz = t(lambda: x.zero_())
    tc = t(lambda: y.copy_(x))
    tr = t(lambda: x.sum())
    ta = t(lambda: x.add_(1.0))
    print(f'{mb:5d} MB: fill {mb/1024/tz/1024*1.0:.2f} TB/s  copy(r+w) {2*mb/1024/1024/tc:.2f} TB/s  read(sum) {mb/1024/1024/tr:.2f} TB/s  rmw(add_) {2*mb/1024/1024/ta:.2f} TB/s')
