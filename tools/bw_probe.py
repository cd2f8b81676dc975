import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for mb in (110, 450, 1800):
    n = mb*1024*1024//2
    a = torch.randn(n, device='cuda', dtype=torch.float16); b = torch.empty_like(a); c = torch.randn_like(a)
    ms = t(lambda: b.copy_(a)); print(mb, 'copy', round(2*n*2/ms/1e9,2), 'TB/s')
    ms = t(lambda: torch.add(a, c, out=b)); print(mb, 'add', round(3*n*2/ms/1e9,2), 'TB/s')
    ms = t(lambda: a.sum()); print(mb, 'sum(read)', round(n*2/ms/1e9,2), 'TB/s')
    ms = t(lambda: b.fill_(1.0)); print(mb, 'fill(write)', round(n*2/ms/1e9,2), 'TB/s')
