"""HBM streaming ceilings with torch's own elementwise kernels (calibration for the HBM-bound rows):
pure write (fill), pure read (sum), copy (1R:1W), add (2R:1W), and a 1R:4W pattern."""
import torch, time
dev = 'cuda'
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (128, 411, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a); c = torch.empty_like(a)
    t = bench(lambda: b.fill_(1.0)); print('%5d MB fill  (W)      %6.1f us %5.2f TB/s' % (mb, t * 1e6, mb * 1.048576e6 / t / 1e12))
    t = bench(lambda: b.copy_(a)); print('%5d MB copy  (1R:1W)  %6.1f us %5.2f TB/s' % (mb, t * 1e6, 2 * mb * 1.048576e6 / t / 1e12))
    t = bench(lambda: torch.add(a, b, out=c)); print('%5d MB add   (2R:1W)  %6.1f us %5.2f TB/s' % (mb, t * 1e6, 3 * mb * 1.048576e6 / t / 1e12))
    af = a.view(torch.int16)
    t = bench(lambda: af.sum()); print('%5d MB sum   (R)      %6.1f us %5.2f TB/s' % (mb, t * 1e6, mb * 1.048576e6 / t / 1e12))
    del a, b, c, af
