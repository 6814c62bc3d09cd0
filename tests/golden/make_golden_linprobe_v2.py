"""Generate tests/golden/lp_*.npz by EXECUTING THE REFERENCE's v2 linear-probe sources on torch-CPU through the paddle
shim: the models (passl/models/simsiam.py SimSiamLinearProbe, passl/models/mocov3.py MoCoV3LinearProbe), the loss and
metric (passl/loss CombinedLoss / CELoss, passl/metric CombinedMetrics / TopkAcc), the update rules
(passl/optimizer/momentum_larc.py, momentum.py over optimizer.py), the schedule (passl/scheduler/lr_scheduler.py
TimmCosine) and the loops themselves (passl/engine/loops/classification_loop.py: ClassificationTrainingEpochLoop.
train_one_step and ClassificationEvaluationLoop.eval_one_dataset) driven by a stand-in trainer object — see
oracle/ref_runner_v2.py for what is replaced (logger, io, profiler, grad_sync: one rank).  backward = torch autograd
over the reference's forward graph.

    python tests/golden/make_golden_linprobe_v2.py          # its own process

Seed-defined inputs: weights oracle.linprobe_v2.LinearProbeOracle(kind, seed=0); per step x ~ N(0,1) [N,3,S,S] and
labels ~ U{0..classes-1} from torch.Generator().manual_seed(4242); two evaluation batches (N and N/2 rows) from the
same generator afterwards.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner_v2                                         # noqa: E402
from oracle import linprobe_v2 as L                                      # noqa: E402
from oracle.mocov3 import SMALL                                          # noqa: E402

VIT_SMALL = {k: v for k, v in SMALL.items()}
CASES = {
    # the SimSiam recipe's optimizer block (MomentumLARC, trust 0.001, no clip, no decay), TimmCosine per STEP here so
    # that three steps see three rates
    'lp_simsiam_r50': dict(kind='simsiam', N=16, S=64, classes=40, steps=3,
                           opt=dict(name='MomentumLARC', momentum=0.9, weight_decay=0.0, trust_coefficient=0.001,
                                    clip=False),
                           sched=dict(learning_rate=1.6, decay_unit='step', epochs=2, step_each_epoch=3, last_epoch=0)),
    # the other branches of the rule: weight decay inside the trust ratio and the clip
    'lp_simsiam_r50_clip': dict(kind='simsiam', N=16, S=64, classes=40, steps=3,
                                opt=dict(name='MomentumLARC', momentum=0.9, weight_decay=1e-3, trust_coefficient=0.02,
                                         clip=True),
                                sched=dict(learning_rate=0.05, decay_unit='step', epochs=2, step_each_epoch=3,
                                           last_epoch=0)),
    # the MoCo-v3 recipe's optimizer block (Momentum, no decay) on a small ViT; warm-up + default last_epoch = -1:
    # the first step runs at get_lr(-1) = warmup_start_lr
    'lp_mocov3_small': dict(kind='mocov3', N=16, S=64, classes=40, steps=4,
                            opt=dict(name='Momentum', momentum=0.9, weight_decay=0.0),
                            sched=dict(learning_rate=0.5, decay_unit='step', epochs=4, step_each_epoch=2,
                                       warmup_epoch=1, warmup_start_lr=0.01)),
}


def build_model(ns, kind, classes):
    if kind == 'simsiam':
        return ns.simsiam.simsiam_resnet50_linearprobe(class_num=classes)
    m = ns.mocov3.MoCoV3LinearProbe(img_size=VIT_SMALL['img_size'], patch_size=VIT_SMALL['patch_size'],
                                    embed_dim=VIT_SMALL['embed_dim'], depth=VIT_SMALL['depth'],
                                    num_heads=VIT_SMALL['num_heads'], mlp_ratio=VIT_SMALL['mlp_ratio'], qkv_bias=True,
                                    class_num=classes, epsilon=1e-6)
    return m


class _Scaler(object):
    """paddle.amp.GradScaler(enable=False)."""

    def scale(self, x):
        return x

    def step(self, opt):
        opt.step()

    def update(self):
        pass


class _Loader(list):
    """A list of batches with the two attributes eval_one_dataset reads."""

    @property
    def dataset(self):
        return range(sum(b[0].shape[0] for b in self))


def run_case(ns, name, kind, N, S, classes, steps, opt, sched):
    torch.manual_seed(0)
    cfg = VIT_SMALL if kind == 'mocov3' else None
    oracle = L.LinearProbeOracle(kind, class_num=classes, seed=0, cfg=cfg)
    model = build_model(ns, kind, classes)
    with torch.no_grad():
        sd = model.state_dict()
        assert set(sd.keys()) == set(oracle.st.keys()), set(sd.keys()) ^ set(oracle.st.keys())
        for k, t in oracle.st.items():
            assert sd[k].shape == t.shape, (k, sd[k].shape, t.shape)
            sd[k].copy_(t)
    train_params = [p for p in model.parameters() if not p.stop_gradient]
    names = {id(p): n for n, p in model.named_parameters()}
    assert sorted(names[id(p)] for p in train_params) == sorted(L.HEAD[kind])
    scheduler = ns.lr_scheduler.TimmCosine(**sched)
    opt = dict(opt)
    klass = getattr(ns.momentum_larc if opt['name'] == 'MomentumLARC' else ns.momentum, opt.pop('name'))
    optimizer = klass(train_params, lr=scheduler, **opt)
    trainer = types.SimpleNamespace(
        model=model, optimizer=optimizer, scaler=_Scaler(), accum_steps=1, fp16=False, fp16_level='O0',
        fp16_custom_white_list=None, fp16_custom_black_list=None, lr_decay_unit=sched['decay_unit'],
        print_batch_step=1, enabled_ema=False, use_dali=False, cur_epoch_id=1, mode='train', validating=True,
        train_loss_func=ns.loss.build_loss([{'CELoss': {'weight': 1.0}}]),
        eval_loss_func=ns.loss.build_loss([{'CELoss': {'weight': 1.0}}]),
        train_metric_func=ns.metric.build_metrics([{'TopkAcc': {'topk': [1, 5]}}]),
        eval_metric_func=ns.metric.build_metrics([{'TopkAcc': {'topk': [1, 5]}}]))
    CL = ns.classification_loop
    train_loop = CL.ClassificationTrainingEpochLoop(trainer, epochs=sched['epochs'])
    eval_loop = CL.ClassificationEvaluationLoop(trainer)
    gen = torch.Generator().manual_seed(4242)
    out = {}
    model.train()
    for s in range(steps):
        x = torch.randn(N, 3, S, S, generator=gen)
        y = torch.randint(0, classes, (N,), generator=gen)
        out['s%d_lr' % s] = np.float64(optimizer.get_lr())
        w0 = {n: p.detach().clone() for n, p in model.named_parameters() if not p.stop_gradient}
        # the step the loop takes (global_step is advanced by train_one_epoch before the call)
        train_loop.global_step += 1
        # gradients are cleared inside train_one_step: keep them through a spy on the optimizer
        grads = {}
        step_fn = optimizer.step

        def spy():
            for n, p in model.named_parameters():
                if not p.stop_gradient:
                    grads[n] = p.grad.detach().clone()
            step_fn()
        optimizer.step = spy
        logits, loss_dict = train_loop.train_one_step([x, y])
        optimizer.step = step_fn
        metric = trainer.train_metric_func(logits, y)
        out['s%d_loss' % s] = np.float64(float(loss_dict['loss']))
        out['s%d_top1' % s], out['s%d_top5' % s] = np.float64(metric['top1']), np.float64(metric['top5'])
        out['s%d_scores_head' % s] = logits.detach()[:, :8].double().numpy().copy()
        for n in L.HEAD[kind]:
            p = dict(model.named_parameters())[n]
            out['s%d_grad/%s' % (s, n)] = grads[n].double().numpy().copy()
            out['s%d_delta/%s' % (s, n)] = (p.detach() - w0[n]).double().numpy().copy()
            out['s%d_pnorm/%s' % (s, n)] = np.float64(p.detach().double().norm().item())
        print(name, 'step', s, 'lr %.6f loss %.6f top1 %.4f top5 %.4f' % (
            out['s%d_lr' % s], out['s%d_loss' % s], out['s%d_top1' % s], out['s%d_top5' % s]))
    # frozen state really is frozen
    sd = model.state_dict()
    for k, t in oracle.st.items():
        if k not in L.HEAD[kind]:
            assert torch.equal(sd[k], t), k
    batches = _Loader()
    for n in (N, N // 2):
        batches.append([torch.randn(n, 3, S, S, generator=gen), torch.randint(0, classes, (n,), generator=gen)])
    res = eval_loop.eval_one_dataset(batches)
    for k, v in res.items():
        out['eval_' + k] = np.float64(float(v))
    print(name, 'eval', {k: round(float(v), 5) for k, v in res.items()})
    out['meta'] = np.array([N, S, classes, steps])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    ns = ref_runner_v2.load_loops(ref_runner_v2.load_solver(ref_runner_v2.load_simsiam()))
    only = sys.argv[1:]
    for name, c in CASES.items():
        if not only or name in only:
            run_case(ns, name, **c)
