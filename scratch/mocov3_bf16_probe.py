"""How does the bf16 gradient error of the MoCo-v3 step (vs the fp32 oracle) depend on the batch size?
(BatchNorm over N rows in the projector / predictor amplifies input rounding ~ 1/sqrt(N).)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import mocov3_util as U
from oracle import mocov3 as O

def relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))

for N in (8, 32, 128):
    for dtype in (torch.float32, torch.bfloat16):
        oracle = O.MoCoV3Oracle(O.SMALL, seed=0, max_steps=10, **U.SOLVER)
        model, opt = U.build_product(O.SMALL, dtype, max_steps=10)
        U.load_oracle_state(model, oracle)
        model.train()
        gen = torch.Generator().manual_seed(5)
        x1 = torch.randn(N, 3, 64, 64, generator=gen); x2 = torch.randn(N, 3, 64, 64, generator=gen)
        cap = {}
        orig = model.contrastive_loss
        def spy(q, k, cap=cap, orig=orig):
            cap.setdefault('q', []).append(q.detach().float().cpu()); cap.setdefault('k', []).append(k.detach().float().cpu())
            return orig(q, k)
        model.contrastive_loss = spy
        loss = U.product_step(model, opt, x1.cuda(), x2.cuda())
        ref = oracle.forward_backward(x1, x2)
        ps = dict(model.named_parameters())
        errs = {n: relmax(ps[n].grad, g) for n, g in ref['grads'].items() if n != 'base_encoder.norm.bias' and not n.endswith('qkv.bias')}
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print('N=%d %s loss %.5f ref %.5f  q1 err %.3e k2 err %.3e  grad err median %.3e worst %s' % (
            N, str(dtype).split('.')[-1], float(loss), float(ref['loss']), relmax(cap['q'][0], ref['q1']), relmax(cap['k'][0], ref['k2']),
            sorted(errs.values())[len(errs)//2], worst), flush=True)
