import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from passl_amd.hip import config, nn
config.set_device('gpu'); config.set_compute_dtype(torch.float32)
gen = torch.Generator().manual_seed(0)
for M, C, relu in ((8, 2048, True), (8, 512, True), (8, 2048, False), (64, 2048, True), (8, 1024, True), (16, 2048, True), (8, 4096, True)):
    bn = nn.BatchNorm1D(C)
    arena = nn.EncoderArena(bn, trainable=True)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=gen) + 0.5); bn.bias.copy_(torch.randn(C, generator=gen) * 0.3)
    arena.refresh()
    x = torch.randn(M, C, generator=gen)
    dy = torch.randn(M, C, generator=gen)
    xr = x.double().requires_grad_(True)
    g = bn.weight.detach().cpu().double().requires_grad_(True); b = bn.bias.detach().cpu().double().requires_grad_(True)
    y = F.batch_norm(xr, None, None, g, b, True, 0.1, 1e-5)
    if relu: y = F.relu(y)
    y.backward(dy.double())
    xd = x.cuda().requires_grad_(True)
    arena.clear_grad()
    z = bn(xd, relu=relu)
    z.backward(dy.cuda())
    torch.cuda.synchronize()
    def rel(a, r): return float((a.double().cpu() - r).abs().max() / r.abs().max())
    print('M=%d C=%d relu=%d  fwd %.2e  dx %.2e  dgamma %.2e  dbeta %.2e' % (M, C, relu, rel(z.detach(), y.detach()), rel(xd.grad, xr.grad), rel(bn.weight.grad, g.grad), rel(bn.bias.grad, b.grad)))
