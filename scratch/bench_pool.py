"""Stem max-pool (3x3 s2 p1) forward / backward at bs 256, 112^2 x 64, bf16."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops
DEV = 'cuda'
xs = [torch.randn(256, 112, 112, 64, device=DEV).bfloat16() for _ in range(3)]
def timeit(f, n=10):
    for _ in range(3): f(0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): f(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
y, idx = ops.maxpool_fwd(xs[0])
dys = [torch.randn_like(y) for _ in range(3)]
tf = timeit(lambda i: ops.maxpool_fwd(xs[i % 3]))
tb = timeit(lambda i: ops.maxpool_bwd(dys[i % 3], idx, 112, 112))
print('maxpool fwd %.1f us (%.2f TB/s of 565 MB)   bwd %.1f us (%.2f TB/s of 565 MB)' % (tf, 565e6 / tf / 1e6, tb, 565e6 / tb / 1e6))
