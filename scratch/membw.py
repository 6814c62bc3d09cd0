import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e-3
for mb in (64, 411, 2048):
    n = mb*1024*1024//2
    a = torch.empty(n, dtype=torch.bfloat16, device='cuda'); b = torch.empty_like(a)
    tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.view(torch.int16).max())
    ta = t(lambda: torch.add(a, a, out=b))
    print('%5d MB: fill %.0f GB/s  copy(r+w) %.0f GB/s  read(max) %.0f GB/s  add(r+w) %.0f GB/s' % (mb, mb/1024/tf, 2*mb/1024/tc, mb/1024/tr, 2*mb/1024/ta))
