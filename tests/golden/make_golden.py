"""Generate the golden vectors in tests/golden/*.npz by EXECUTING THE REFERENCE.

Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

For each case the reference's own MoCo (passl_v110/modeling/architectures/moco.py
built through its registries, see oracle/ref_runner.py) runs ``steps`` training
iterations on torch-CPU through the paddle shim; the backward pass is torch
autograd over the reference's forward graph and the Momentum update is
oracle.moco.MoCoOracle.apply_momentum (Paddle's optimizer kernel is not in the
reference tree).  Inputs and initial weights are seed-defined so that the GPU
box can regenerate them without /root/reference:

    weights/queue : oracle.moco.MoCoOracle(seed=0, K=K)
    views         : torch.Generator().manual_seed(1234); per step x_q then x_k ~ N(0,1)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_runner                      # noqa: E402
from oracle.moco import MoCoOracle                 # noqa: E402

CASES = {
    # BASELINE.json configs[0]: MoCo-v2 R50, 2x224^2 views, bs=32, K=65536, 1 rank
    'moco_v2_r50_cfg1': dict(N=32, hw=224, K=65536, steps=2),
    # fast case for every-commit runs
    'moco_v2_r50_small': dict(N=8, hw=64, K=1024, steps=3),
    # configs/moco/moco_v1_r50.yaml: LinearNeck, T = 0.07, lr 0.03 MultiStepDecay
    'moco_v1_r50_small': dict(N=8, hw=64, K=1024, steps=3, v1=True),
    # BASELINE.json configs[1] (the benchmarked shape): bs=256, 2x224^2, K=65536; ONE step of the
    # reference (a float64 re-evaluation at this size does not fit the build container's memory)
    'moco_v2_r50_cfg2': dict(N=256, hw=224, K=65536, steps=1, f64=False),
}
V1 = dict(neck='LinearNeck', T=0.07, lr=0.03, milestones=[120 * 5004, 160 * 5004])
WATCH = ['0.conv1.weight', '0.layer1.0.conv2.weight', '0.layer2.0.downsample.0.weight',
         '0.layer4.2.conv3.weight', '0.layer3.5.bn2.weight', '0.bn1.bias',
         '1.mlp.0.weight', '1.mlp.2.weight', '1.mlp.2.bias']
WATCH_STATS = ['0.bn1._mean', '0.bn1._variance', '0.layer4.2.bn3._mean',
               '0.layer4.2.bn3._variance']


def views(gen, N, hw):
    xq = torch.randn(N, 3, hw, hw, generator=gen)
    xk = torch.randn(N, 3, hw, hw, generator=gen)
    return xq, xk


def run_case(name, N, hw, K, steps, v1=False, f64=True):
    torch.manual_seed(0)
    okw = dict(V1) if v1 else {}
    watch = [n for n in WATCH if not n.startswith('1.')] + (['1.fc.weight', '1.fc.bias'] if v1 else
                                                            [n for n in WATCH if n.startswith('1.')])
    oracle = MoCoOracle(K=K, seed=0, t_max=200 * 5004, **okw)
    model = ref_runner.build_reference_moco(K=K, T=okw.get('T', 0.2), neck=okw.get('neck', 'NonLinearNeckV1'))
    ref_runner.load_oracle_state(model, oracle)
    model.train()
    captured = {}
    model.head.register_forward_pre_hook(
        lambda mod, args: captured.update(pos=args[0].detach().clone(),
                                          neg=args[1].detach().clone()))
    gen = torch.Generator().manual_seed(1234)
    out = {}
    for s in range(steps):
        xq, xk = views(gen, N, hw)
        ptr0 = int(model.queue_ptr[0])
        for p in model.parameters():
            p.grad = None
        res = model(xq, xk, mode='train', total_iters=steps, current_iter=s + 1, mixup_fn=None)
        res['loss'].backward()
        qsd = dict(model.encoder_q.named_parameters())
        grads = {n: qsd[n].grad.detach().clone() for n in qsd if qsd[n].requires_grad}
        # Momentum step (restated) applied to the reference model's own parameters
        oracle.q = {n: p.detach().clone() for n, p in model.encoder_q.state_dict().items()}
        oracle.apply_momentum(grads)
        with torch.no_grad():
            for n, p in model.encoder_q.state_dict().items():
                p.copy_(oracle.q[n])
        ksd = model.encoder_k.state_dict()
        pre = 's%d_' % s
        out[pre + 'loss'] = np.float64(res['loss'].item())
        out[pre + 'acc1'] = np.float64(float(res['acc1']))
        out[pre + 'acc5'] = np.float64(float(res['acc5']))
        logits = torch.cat((captured['pos'], captured['neg']), dim=1) / model.head.temperature
        out[pre + 'logits_head'] = logits[:, :8].numpy().copy()
        out[pre + 'logits_rowsum64'] = logits.double().sum(dim=1).numpy()
        out[pre + 'logits_rowlse64'] = torch.logsumexp(logits.double(), dim=1).numpy()
        out[pre + 'queue_ptr'] = np.int64(int(model.queue_ptr[0]))
        out[pre + 'queue_new'] = model.queue[:, ptr0:ptr0 + N].detach().numpy().copy()
        out[pre + 'queue_sum64'] = np.float64(model.queue.double().sum().item())
        for n in watch:
            out[pre + 'gradnorm/' + n] = np.float64(grads[n].double().norm().item())
            out[pre + 'qnorm/' + n] = np.float64(oracle.q[n].double().norm().item())
            out[pre + 'knorm/' + n] = np.float64(ksd[n].double().norm().item())
            out[pre + 'kdot/' + n] = np.float64(
                (ksd[n].double() * oracle.q[n].double()).sum().item())
        for n in WATCH_STATS:
            out[pre + 'qstat/' + n] = oracle.q[n][:8].numpy().astype(np.float64)
            out[pre + 'kstat/' + n] = ksd[n][:8].numpy().astype(np.float64)
        print(name, 'step', s, 'loss %.6f acc1 %.2f acc5 %.2f ptr %d' % (
            out[pre + 'loss'], out[pre + 'acc1'], out[pre + 'acc5'], out[pre + 'queue_ptr']))
    # The same steps evaluated in float64 (oracle, identical init and inputs).  A random-init
    # R50 with batch-stat BN is ill-conditioned: fp32 and fp64 evaluations of the SAME algorithm
    # drift apart after the first update, so the tests bound |HIP - ref32| by a small multiple
    # of |ref32 - ref64| where that exceeds the nominal 1e-3.
    del model, res, grads
    if not f64:
        steps_f64 = 0
    else:
        steps_f64 = steps
    o64 = MoCoOracle(K=K if f64 else 128, seed=0, t_max=200 * 5004, **okw)
    for d in (o64.q, o64.k):
        for n in d:
            d[n] = d[n].double()
    o64.queue = o64.queue.double()
    gen = torch.Generator().manual_seed(1234)
    for s in range(steps_f64):
        xq, xk = views(gen, N, hw)
        ptr0 = o64.queue_ptr
        r = o64.train_step(xq.double(), xk.double())
        pre = 's%d_f64_' % s
        out[pre + 'loss'] = np.float64(float(r['loss']))
        out[pre + 'logits_head'] = r['logits'][:, :8].numpy().copy()
        out[pre + 'queue_new'] = o64.queue[:, ptr0:ptr0 + N].numpy().copy()
        for n in watch:
            out[pre + 'gradnorm/' + n] = np.float64(r['grads'][n].norm().item())
            out[pre + 'qnorm/' + n] = np.float64(o64.q[n].norm().item())
            out[pre + 'knorm/' + n] = np.float64(o64.k[n].norm().item())
        for n in WATCH_STATS:
            out[pre + 'qstat/' + n] = o64.q[n][:8].numpy().copy()
            out[pre + 'kstat/' + n] = o64.k[n][:8].numpy().copy()
        print(name, 'f64 step', s, 'loss %.6f' % out[pre + 'loss'])
    del o64
    if not v1:
        out.update(bf16_steps(N, hw, K, steps, watch))
    out['meta'] = np.array([N, hw, K, steps], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def bf16_steps(N, hw, K, steps, watch):
    """The same steps in the oracle's bf16-EMULATING mode (oracle/bf16.py): the reference's fp32
    algorithm with round-to-nearest-even bfloat16 at the tensors the MI355X path stores in bf16.  The
    reference has no bf16 MoCo path; these `s<k>_bf16_*` entries are what the product's bf16 mode is
    held to (a tight bound) next to the fp32 reference values above (a loose sanity bound)."""
    out = {}
    # variant 'bf16': fp32 accumulation; 'bf16b': float64 accumulation of the same bf16 products — a second
    # VALID evaluation of the same contract (only the FIRST step: it measures how ill-conditioned the
    # step is with respect to summation order; the tests scale their bounds with |bf16 - bf16b|)
    from oracle import resnet50 as R50
    for tag, acc64, nsteps in (('bf16', False, steps), ('bf16b', True, 1)):
        R50.ACCUM64 = acc64
        try:
            out.update(_bf16_variant(tag, N, hw, K, nsteps, watch))
        finally:
            R50.ACCUM64 = False
    return out


def _bf16_variant(tag, N, hw, K, steps, watch):
    out = {}
    ob = MoCoOracle(K=K, seed=0, t_max=200 * 5004, bf16=True)
    gen = torch.Generator().manual_seed(1234)
    for s in range(steps):
        xq, xk = views(gen, N, hw)
        ptr0 = ob.queue_ptr
        r = ob.train_step(xq, xk)
        pre = 's%d_%s_' % (s, tag)
        out[pre + 'loss'] = np.float64(float(r['loss']))
        out[pre + 'acc1'] = np.float64(float(r['acc1']))
        out[pre + 'acc5'] = np.float64(float(r['acc5']))
        out[pre + 'logits_head'] = r['logits'][:, :8].numpy().copy()
        out[pre + 'logits_rowlse64'] = torch.logsumexp(r['logits'].double(), dim=1).numpy()
        out[pre + 'queue_new'] = ob.queue[:, ptr0:ptr0 + N].numpy().copy()
        for n in watch:
            out[pre + 'gradnorm/' + n] = np.float64(r['grads'][n].double().norm().item())
            out[pre + 'qnorm/' + n] = np.float64(ob.q[n].double().norm().item())
            out[pre + 'knorm/' + n] = np.float64(ob.k[n].double().norm().item())
        for n in WATCH_STATS:
            out[pre + 'qstat/' + n] = ob.q[n][:8].numpy().astype(np.float64)
            out[pre + 'kstat/' + n] = ob.k[n][:8].numpy().astype(np.float64)
        print(tag, 'emulated step', s, 'loss %.6f' % out[pre + 'loss'], flush=True)
        del r
    return out


def add_bf16(name, N, hw, K, steps, v1=False, f64=True):
    """Add the bf16-emulated entries to an existing golden file without re-running the reference."""
    path = os.path.join(HERE, name + '.npz')
    z = dict(np.load(path))
    z.update(bf16_steps(N, hw, K, steps, WATCH))
    np.savez_compressed(path, **z)


if __name__ == '__main__':
    # python make_golden.py [case ...]            run the reference (+ f64 + bf16-emulated oracle)
    # python make_golden.py --add-bf16 [case ...]  only (re)compute the bf16-emulated entries
    args = sys.argv[1:]
    if args and args[0] == '--add-bf16':
        for name in (args[1:] or [n for n in CASES if not CASES[n].get('v1')]):
            add_bf16(name, **CASES[name])
        sys.exit(0)
    assert ref_runner.available(), 'needs /root/reference'
    which = args or list(CASES)
    for name in which:
        run_case(name, **CASES[name])
