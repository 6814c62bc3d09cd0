"""Generate tests/golden/mocov3_*.npz by EXECUTING THE REFERENCE's v2 MoCo-v3 sources
(passl/models/mocov3.py MoCoV3ViT + MoCoV3Pretrain, passl/models/vision_transformer.py,
passl/models/utils/averaged_model.py CosineEMA) on torch-CPU through the paddle shim (oracle/ref_runner_v2.py);
backward = torch autograd over the reference's forward graph, AdamW = oracle.mocov3.MoCoV3Oracle.apply_adamw
(Paddle's optimizer kernel is not in the reference tree).

    python tests/golden/make_golden_mocov3.py          # its own process: see oracle/ref_runner_v2.py

Seed-defined inputs (regenerable without /root/reference):
    weights: oracle.mocov3.MoCoV3Oracle(cfg, seed=0);  per step: x1, x2 ~ N(0,1) from
    torch.Generator().manual_seed(777) (rank r of the two-rank case: manual_seed(777 + r))

The two-rank case runs the reference's forward once per rank with paddle.distributed.{get_world_size, get_rank,
all_gather} answering as rank r of 2: the keys of the OTHER rank (no gradient flows through them) come from a first
pass over a deep copy of that rank's model.
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner_v2                                   # noqa: E402
from oracle.mocov3 import MoCoV3Oracle, VIT_B, SMALL, is_buffer    # noqa: E402

CASES = {
    'mocov3_small': dict(cfg=SMALL, N=8, steps=3, max_steps=10),
    # tasks/ssl/mocov3/configs/mocov3_vit_base_patch16_224_pt_in1k_4n32c_dp_fp16o1.yaml architecture, tiny batch
    'mocov3_vit_b': dict(cfg=VIT_B, N=4, steps=2, max_steps=100),
}
SOLVER = dict(lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.1)
WATCH = ['base_encoder.cls_token', 'base_encoder.blocks.0.attn.qkv.weight', 'base_encoder.blocks.1.mlp.fc2.bias',
         'base_encoder.blocks.1.norm2.weight', 'base_encoder.norm.bias', 'base_encoder.head.0.weight',
         'base_encoder.head.4.bias', 'base_encoder.head.6.weight', 'predictor.0.weight', 'predictor.1.weight',
         'predictor.3.weight']
WATCH_MOM = ['base_encoder.patch_embed.proj.weight', 'base_encoder.blocks.0.attn.qkv.weight',
             'base_encoder.head.6.weight', 'predictor.3.weight', 'base_encoder.head.7._mean',
             'predictor.1._variance']


def ref_key(k, momentum=False):
    if not momentum:
        return k
    if k.startswith('base_encoder.'):
        return 'momentum_encoder.model.0.' + k[len('base_encoder.'):]
    return 'momentum_encoder.model.1.' + k[len('predictor.'):]


def build_reference(ns, cfg, T, base_momentum, max_steps):
    from functools import partial
    paddle = sys.modules['paddle']
    ns.runtime_info_hub.max_steps = max_steps
    enc = partial(ns.mocov3.MoCoV3ViT, stop_grad_conv1=True, img_size=cfg['img_size'], patch_size=cfg['patch_size'],
                  embed_dim=cfg['embed_dim'], depth=cfg['depth'], num_heads=cfg['num_heads'],
                  mlp_ratio=cfg['mlp_ratio'], qkv_bias=True,
                  norm_layer=partial(paddle.nn.LayerNorm, epsilon=1e-6))
    return ns.mocov3.MoCoV3Pretrain(base_encoder=enc, dim=cfg['dim'], mlp_dim=cfg['mlp_dim'], T=T,
                                    base_momentum=base_momentum)


def load_state(model, oracle):
    with torch.no_grad():
        sd = model.state_dict()
        want = set(oracle.st) | {ref_key(k, True) for k in oracle.mom} | {'momentum_encoder.steps'}
        assert set(sd.keys()) == want, set(sd.keys()) ^ want
        for k, t in oracle.st.items():
            assert sd[k].shape == t.shape, (k, sd[k].shape, t.shape)
            sd[k].copy_(t.detach())
        for k, t in oracle.mom.items():
            sd[ref_key(k, True)].copy_(t.detach())


def pull_state(model, oracle):
    sd = model.state_dict()
    for k in oracle.st:
        oracle.st[k] = sd[k].detach().clone()
    for k in oracle.mom:
        oracle.mom[k] = sd[ref_key(k, True)].detach().clone()
    oracle.steps = int(sd['momentum_encoder.steps'])


def record(out, pre, loss, grads, oracle):
    out[pre + 'loss'] = np.float64(float(loss))
    for n in WATCH:
        out[pre + 'gradnorm/' + n] = np.float64(grads[n].double().norm().item())
        out[pre + 'pnorm/' + n] = np.float64(oracle.st[n].double().norm().item())
    for n in WATCH_MOM:
        out[pre + 'mom_pnorm/' + n] = np.float64(oracle.mom[n].double().norm().item())


def run_case(ns, name, cfg, N, steps, max_steps):
    torch.manual_seed(0)
    oracle = MoCoV3Oracle(cfg, seed=0, max_steps=max_steps, **SOLVER)
    model = build_reference(ns, cfg, 0.2, 0.99, max_steps)
    load_state(model, oracle)
    model.train()
    gen = torch.Generator().manual_seed(777)
    out, S = {}, cfg['img_size']
    for s in range(steps):
        x1 = torch.randn(N, 3, S, S, generator=gen)
        x2 = torch.randn(N, 3, S, S, generator=gen)
        for p in model.parameters():
            p.grad = None
        loss = model([x1, x2])
        loss.backward()
        ps = dict(model.named_parameters())
        grads = {n: ps[n].grad.detach().clone() for n in ps if ps[n].grad is not None}
        pull_state(model, oracle)
        from oracle.mocov3 import trainable_keys
        assert set(grads) == set(trainable_keys(oracle.st)), set(grads) ^ set(trainable_keys(oracle.st))
        oracle.apply_adamw(grads)
        with torch.no_grad():
            sd = model.state_dict()
            for n in oracle.st:
                sd[n].copy_(oracle.st[n])
        pre = 's%d_' % s
        record(out, pre, loss.item(), grads, oracle)
        out[pre + 'ema_steps'] = np.int64(oracle.steps)
        print(name, 'step', s, 'loss %.6f' % out[pre + 'loss'])
    # the fp64 trajectory of the restatement on the same inputs: the tolerance anchor of the GPU tests
    o64 = MoCoV3Oracle(cfg, seed=0, max_steps=max_steps, dtype=torch.float64, **SOLVER)
    gen = torch.Generator().manual_seed(777)
    for s in range(steps):
        x1 = torch.randn(N, 3, S, S, generator=gen)
        x2 = torch.randn(N, 3, S, S, generator=gen)
        r = o64.train_step(x1.double(), x2.double())
        pre = 's%d_f64_' % s
        record(out, pre, r['loss'], r['grads'], o64)
        if s == 0:
            out['s0_f64_q1_head'] = r['q1'][:, :8].numpy().copy()
            out['s0_f64_k2_head'] = r['k2'][:, :8].numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def run_two_ranks(ns, name='mocov3_small_2rank', cfg=SMALL, N=4, max_steps=10):
    """One step of two data-parallel ranks, loss and (un-reduced) gradient norms of each rank."""
    paddle = sys.modules['paddle']
    dist = paddle.distributed
    saved = dist.get_world_size, dist.get_rank, dist.all_gather
    out, S = {}, cfg['img_size']
    oracle = MoCoV3Oracle(cfg, seed=0, max_steps=max_steps, **SOLVER)
    models, xs = [], []
    for r in range(2):
        torch.manual_seed(0)
        m = build_reference(ns, cfg, 0.2, 0.99, max_steps)
        load_state(m, oracle)
        m.train()
        gen = torch.Generator().manual_seed(777 + r)
        xs.append((torch.randn(N, 3, S, S, generator=gen), torch.randn(N, 3, S, S, generator=gen)))
        models.append(m)
    try:
        # pass 1: every rank's keys (k2 is gathered first, then k1: contrastive_loss(q1, k2) + contrastive_loss(q2, k1))
        keys = []
        for r in range(2):
            got = []
            dist.get_world_size = lambda group=None: 2
            dist.get_rank = lambda group=None, r=r: r
            dist.all_gather = lambda lst, t, group=None, got=got: (got.append(t.detach().clone()),
                                                                    lst.extend([t, t]))[0]
            copy.deepcopy(models[r])(list(xs[r]))
            assert len(got) == 2
            keys.append(got)
        for r in range(2):
            call = [0]

            def gather(lst, t, group=None, r=r, call=call):
                i = call[0]
                call[0] += 1
                assert torch.equal(t, keys[r][i])
                lst.extend([keys[0][i], keys[1][i]])
            dist.get_rank = lambda group=None, r=r: r
            dist.all_gather = gather
            loss = models[r](list(xs[r]))
            loss.backward()
            ps = dict(models[r].named_parameters())
            grads = {n: ps[n].grad.detach().clone() for n in ps if ps[n].grad is not None}
            pre = 'r%d_' % r
            out[pre + 'loss'] = np.float64(loss.item())
            for n in WATCH:
                out[pre + 'gradnorm/' + n] = np.float64(grads[n].double().norm().item())
            print(name, 'rank', r, 'loss %.6f' % out[pre + 'loss'])
    finally:
        dist.get_world_size, dist.get_rank, dist.all_gather = saved
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    ns = ref_runner_v2.load()
    only = sys.argv[1:]
    for name, c in CASES.items():
        if not only or name in only:
            run_case(ns, name, **c)
    if not only or 'mocov3_small_2rank' in only:
        run_two_ranks(ns)
