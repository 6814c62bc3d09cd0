import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import moco_util as U
from oracle.moco import MoCoOracle
oracle = MoCoOracle(K=512, seed=3, t_max=1000)
gen = torch.Generator().manual_seed(1)
xq = torch.randn(8, 3, 64, 64, generator=gen).cuda(); xk = torch.randn(8, 3, 64, 64, generator=gen).cuda()
res = {}
for mode in ('plain', 'overlap'):
    model, opt, sched = U.build_product(512, torch.float32)
    U.load_oracle_state(model, oracle)
    model.train()
    if mode == 'plain':
        model._key_groups = False
    cap = {}
    orig = model.head.fused
    def spy(q, k, queue, cap=cap, orig=orig):
        cap['q'], cap['k'] = q.detach().clone(), k.detach().clone()
        return orig(q, k, queue)
    model.head.fused = spy
    out = model(xq, xk)
    torch.cuda.synchronize()
    res[mode] = (cap['q'].cpu(), cap['k'].cpu(), model.arena_k.flat.detach().cpu().clone(), model.arena_k.bn_affine[0].cpu().clone(), model.arena_k.bn_affine[1].cpu().clone(), float(out['loss']))
    print(mode, 'overlap used:', bool(model._key_groups), 'loss', float(out['loss']))
a, b = res['plain'], res['overlap']
for i, n in enumerate(('q', 'k', 'arena_k.flat', 'bn_scale', 'bn_shift')):
    d = (a[i] - b[i]).abs()
    print(n, 'max abs diff %.3e' % float(d.max()), 'first bad index', int(d.argmax()) if float(d.max()) > 0 else -1, 'numel', d.numel())
nt = None
# is the staged frozen path itself right?  same state, main stream, no overlap
model, opt, sched = U.build_product(512, torch.float32)
U.load_oracle_state(model, oracle)
model.train()
with torch.no_grad():
    bb, neck = model.encoder_k[0], model.encoder_k[1]
    k_ref = neck(bb(xk))
    y = xk
    for i in range(len(bb.units())):
        y = bb.frozen_unit(i, y)
    k_st = neck(y)
    torch.cuda.synchronize()
    print('staged-vs-forward on one stream: max abs diff %.3e' % float((k_ref - k_st).abs().max()))
    # staged on the key stream with a staged input
    from passl_amd.hip import streams
    key = streams.key_stream(xk.device)
    xs = bb.stage_input(xk)
    main = torch.cuda.current_stream()
    ev = main.record_event()
    with torch.cuda.stream(key):
        key.wait_event(ev)
        y = xs
        for i in range(len(bb.units())):
            y = bb.frozen_unit(i, y)
        k_key = neck(y)
    main.wait_stream(key)
    torch.cuda.synchronize()
    print('staged on key stream: max abs diff %.3e' % float((k_ref - k_key).abs().max()))
