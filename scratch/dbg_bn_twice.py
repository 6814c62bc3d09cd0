import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, torch.nn.functional as F
from passl_amd.hip import config, nn
config.set_device('gpu'); config.set_compute_dtype(torch.float32)
gen = torch.Generator().manual_seed(0)
def rel(a, r): return float((a.double().cpu() - r).abs().max() / r.abs().max())
for C, with_lin in ((512, False), (512, True), (2048, True)):
    M = 8
    bn = nn.BatchNorm1D(C)
    lin = nn.Linear(C, C, bias_attr=False)
    seq = torch.nn.ModuleList([lin, bn])
    arena = nn.EncoderArena(seq, trainable=True)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=gen) + 0.5); bn.bias.copy_(torch.randn(C, generator=gen) * 0.3)
        lin.weight.copy_(torch.randn(C, C, generator=gen) / C ** 0.5)
    arena.refresh()
    xs = [torch.randn(M, C, generator=gen) for _ in range(2)]
    dys = [torch.randn(M, C, generator=gen) for _ in range(2)]
    g = bn.weight.detach().cpu().double().requires_grad_(True); b = bn.bias.detach().cpu().double().requires_grad_(True)
    W = lin.weight.detach().cpu().double().requires_grad_(True)
    xr = [x.double().requires_grad_(True) for x in xs]
    tot = 0
    for x, dy in zip(xr, dys):
        h = x @ W if with_lin else x
        y = F.relu(F.batch_norm(h, None, None, g, b, True, 0.1, 1e-5))
        tot = tot + (y * dy.double()).sum()
    tot.backward()
    xd = [x.cuda().requires_grad_(True) for x in xs]
    arena.clear_grad()
    tot = 0
    for x, dy in zip(xd, dys):
        h = lin(x, out_f32=True) if with_lin else x
        z = bn(h, relu=True)
        tot = tot + (z * dy.cuda()).sum()
    tot.backward()
    torch.cuda.synchronize()
    print('C=%d lin=%d  dx1 %.2e dx2 %.2e dgamma %.2e dbeta %.2e dW %.2e' % (C, with_lin, rel(xd[0].grad, xr[0].grad), rel(xd[1].grad, xr[1].grad), rel(bn.weight.grad, g.grad), rel(bn.bias.grad, b.grad), rel(lin.weight.grad, W.grad) if with_lin else 0))
