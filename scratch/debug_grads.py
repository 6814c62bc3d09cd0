import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import moco_util as U
from oracle.moco import MoCoOracle
K, N = 512, 8
oracle = MoCoOracle(K=K, seed=3, t_max=1000)
model, opt, sched = U.build_product(K, torch.float32)
sched.T_max = 1000
U.load_oracle_state(model, oracle)
model.train()
gen = torch.Generator().manual_seed(99)
for s in range(2):
    xq = torch.randn(N, 3, 96, 80, generator=gen); xk = torch.randn(N, 3, 96, 80, generator=gen)
    taps = {}
    q_before = {n: t.detach().clone() for n, t in oracle.q.items()}
    ref = oracle.train_step(xq, xk, taps=taps)
    out = model(xq.cuda(), xk.cuda())
    opt.clear_grad(); out['loss'].backward()
    qsd = dict(model.encoder_q.named_parameters())
    rows = []
    for n, g in ref['grads'].items():
        gp = qsd[n].grad.cpu()
        d = (gp - g).norm().item() / max(g.norm().item(), 1e-12)
        rows.append((d, n, g.norm().item(), gp.norm().item()))
    rows.sort(reverse=True)
    print('step', s, 'loss', float(out['loss']), float(ref['loss']))
    for r in rows[:25]: print('  %.3e %-40s ref %.4e got %.4e' % r)
    print('  median rel err', sorted(r[0] for r in rows)[len(rows)//2])
    opt.step(); sched.step()
    # param compare after update
    prow = []
    for n in ref['grads']:
        d = (qsd[n].detach().cpu() - oracle.q[n]).norm().item() / max((oracle.q[n]-q_before[n]).norm().item(), 1e-12)
        prow.append((d, n))
    prow.sort(reverse=True)
    print('  worst param-update rel err (relative to update size):', prow[:5])
