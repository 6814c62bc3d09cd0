import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
from passl_amd.hip import ops
from test_simclr_gpu import _head_general, _unit
DEV='cuda'
for (B, BL, roff, T, sa, sb) in [(16,16,0,0.1,1,1),(16,32,0,0.1,1,1),(16,32,16,0.1,1,1),(24,72,24,0.2,1,1),(16,16,0,0.2,1.7,0.6),(16,16,0,0.1,1.0,0.6)]:
    gen = torch.Generator().manual_seed(7)
    a_all = (_unit(BL, gen) * sa).double(); b_all = (_unit(BL, gen) * sb).double()
    h1 = a_all[roff:roff+B].clone().requires_grad_(True); h2 = b_all[roff:roff+B].clone().requires_grad_(True)
    A = a_all.clone().requires_grad_(True); Bm = b_all.clone().requires_grad_(True)
    loss, acc = _head_general(h1, h2, A, Bm, roff, T); loss.backward()
    f = lambda t: t.detach().float().to(DEV).contiguous()
    out, rs = ops.ntxent_fwd(f(h1), f(h2), f(A), f(Bm), roff, T, 3.0)
    da, db, dA, dB = ops.ntxent_bwd(f(h1), f(h2), f(A), f(Bm), rs, None, roff, T, 3.0)
    errs = [float((g.cpu().double()-r).abs().max()/r.abs().max()) for g, r in ((da,h1.grad),(db,h2.grad),(dA,A.grad),(dB,Bm.grad))]
    tot = float(((da.cpu().double()+dA.cpu().double()[roff:roff+B]) - (h1.grad + A.grad[roff:roff+B])).abs().max())
    print((B,BL,roff,T,sa,sb), 'loss err %.2e' % abs(float(out[0])-float(loss)), 'rel errs da db dA dB: %.2e %.2e %.2e %.2e' % tuple(errs), 'sum err %.2e' % tot)
