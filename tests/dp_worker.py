"""Worker of tests/test_dp_gpu.py: one data-parallel rank of a short training run on the HIP path.
Launched by torch.distributed.run; several ranks may share one GPU (PASSL_DEVICE_INDEX) with the
gloo backend (PASSL_DIST_BACKEND) — the collectives then stage through the host, everything else
(kernels, reducer bucketing, broadcast at start-up, gathered keys / embeddings) is the production
path.  Prints `DP-OK <workload> <loss>` on rank 0 when every check passed."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OVERRIDES = {
    'moco': ('configs/moco/moco_v2_r50_synthetic.yaml',
             ['dataloader.train.sampler.batch_size=8', 'dataloader.train.dataset.image_size=64',
              'dataloader.train.dataset.num_samples=256']),
    'simclr': ('configs/simclr/simclr_r50_synthetic.yaml',
               ['dataloader.train.sampler.batch_size=8', 'dataloader.train.dataset.image_size=64',
                'dataloader.train.dataset.num_samples=256']),
    'mae': ('configs/mae/mae_vit_b_synthetic.yaml',
            ['dataloader.train.sampler.batch_size=8', 'dataloader.train.dataset.image_size=64',
             'dataloader.train.dataset.num_samples=256', 'model.architecture.depth=2',
             'model.architecture.embed_dim=128', 'model.architecture.num_heads=4',
             'model.architecture.decoder_embed_dim=64', 'model.architecture.decoder_depth=1',
             'model.architecture.decoder_num_heads=2']),
    'linprobe': ('configs/moco/moco_clas_r50_synthetic.yaml',
                 ['dataloader.train.sampler.batch_size=8', 'dataloader.train.dataset.image_size=64',
                  'dataloader.train.dataset.num_samples=256', 'dataloader.train.dataset.num_classes=16',
                  'lr_scheduler.learning_rate=0.00002']),
    'clip': ('configs/clip/vit-b-32_synthetic.yaml',
             ['dataloader.train.sampler.batch_size=8', 'dataloader.train.dataset.image_size=64',
              'dataloader.train.dataset.num_samples=256', 'dataloader.train.dataset.context_length=16',
              'dataloader.train.dataset.vocab_size=400', 'model.architecture.image_resolution=64',
              'model.architecture.vision_layers=2', 'model.architecture.vision_width=128',
              'model.architecture.context_length=16', 'model.architecture.vocab_size=400',
              'model.architecture.transformer_width=128', 'model.architecture.transformer_heads=2',
              'model.architecture.transformer_layers=2', 'model.architecture.embed_dim=64']),
}


def three_steps(tr):
    from passl_amd.hooks import OptimizerHook, LRSchedulerHook
    opt_hook = next(h for h in tr.hooks if isinstance(h, OptimizerHook))
    lr_hook = next(h for h in tr.hooks if isinstance(h, LRSchedulerHook))
    data = next(iter(tr.train_dataloader))
    losses = []
    for _ in range(3):
        tr.current_iter += 1
        tr.outputs = tr.model(*data, total_iters=tr.total_iters, current_iter=tr.current_iter, mixup_fn=None)
        opt_hook.train_iter_end(tr)
        lr_hook.train_iter_end(tr)
        losses.append(tr.outputs['loss'].detach().clone())
    torch.cuda.synchronize()
    return [float(l) for l in losses]


def plain_run(tr, workload):
    tr.mode = 'train'
    tr.model.train()
    losses = three_steps(tr)
    p1 = torch.cat([p.detach().reshape(-1).double() for p in tr.model.parameters()])
    print('DP-OK %s %.6f digest=%.17g losses=%s' % (workload, losses[-1], float(p1.sum()),
                                                   ','.join('%.9g' % l for l in losses)), flush=True)


OVERRIDES['clipx'] = OVERRIDES['clip']        # + multi_rank: cross-rank InfoNCE (BASELINE configs[4])
OVERRIDES['moco_shuffle'] = OVERRIDES['moco']  # + shuffle_bn: the cross-rank batch shuffle of moco.py:107-152


def mocov3_run():
    """Two data-parallel ranks of MoCo-v3 (small ViT) from the golden case's state and inputs
    (tests/golden/mocov3_small_2rank.npz: the reference's own forward run as rank r of 2): the rank-local loss and
    gradient norms must equal the reference's — which they only do if the keys are gathered in rank order and row
    i's positive is column N*rank + i — then two more steps through the overlapped bucketed all-reduce keep the
    replicas bit-identical."""
    import numpy as np
    import mocov3_util as U
    from oracle import mocov3 as O
    from passl_amd.core.sync_utils import GradReducer, grad_sync, param_sync
    from passl_amd.engine.trainer import _init_distributed
    from passl_amd.hip import config as hip_config
    dev = hip_config.set_device('gpu')
    rank, world = _init_distributed(dev)
    assert world == 2
    cfg, N = O.SMALL, 4
    oracle = O.MoCoV3Oracle(cfg, seed=0, max_steps=10, **U.SOLVER)
    model, opt = U.build_product(cfg, torch.float32, max_steps=10)
    U.load_oracle_state(model, oracle)
    param_sync(model)
    model.train()
    gen = torch.Generator().manual_seed(777 + rank)
    x1 = torch.randn(N, 3, 64, 64, generator=gen).cuda()
    x2 = torch.randn(N, 3, 64, 64, generator=gen).cuda()
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'mocov3_small_2rank.npz'))
    loss = model([x1, x2])
    opt.clear_grad()
    loss.backward()                                   # no reducer yet: the gradients are this rank's own
    torch.cuda.synchronize()
    got = float(loss.detach())
    assert abs(got - float(z['r%d_loss' % rank])) < 2e-5, (rank, got, float(z['r%d_loss' % rank]))
    ps = dict(model.named_parameters())
    for key in z.files:
        if key.startswith('r%d_gradnorm/' % rank) and not key.endswith('norm.bias'):
            n = key.split('/', 1)[1]
            g = ps[n].grad.double().norm().item()
            assert abs(g - float(z[key])) <= 5e-4 * g, (rank, n, g, float(z[key]))
    grad_sync([{'params': opt._parameter_list}])      # blocking all-reduce (mean), as the reference's loop
    opt.step()
    reducer = GradReducer(model.arena_q, opt)
    losses = [got]
    for _ in range(2):
        loss = model([x1, x2])
        opt.clear_grad()
        reducer.begin()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    # (the weights: BatchNorm running statistics stay rank-local, as in the reference's DataParallel)
    nt = model.arena_q.n_train
    for name, t in (('arena_q', model.arena_q.flat[:nt]), ('arena_k', model.arena_k.flat[:nt])):
        ref = t.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, t), '%s differs between ranks (max abs diff %.3e)' % (name, float((ref - t).abs().max()))
    assert model.momentum_encoder._steps == 3
    dist.barrier()
    if rank == 0:
        print('DP-OK mocov3 %.6f losses=%s' % (losses[-1], ','.join('%.9g' % l for l in losses)), flush=True)
    dist.destroy_process_group()


def simsiam_run():
    """Two data-parallel ranks of SimSiam with SyncBatchNorm (reference passl/models/simsiam.py:160-162): every
    BatchNorm normalises over BOTH ranks' batches, so the run must equal the ONE-process step on the concatenated
    batch — loss (mean of the rank losses), averaged gradients, updated parameters and running statistics are
    compared with the fp64 restatement on [x_rank0 ; x_rank1], and the replicas must end bit-identical, statistics
    included."""
    import simsiam_util as U
    from oracle import simsiam as S
    from passl_amd.core.sync_utils import grad_sync, param_sync
    from passl_amd.engine.trainer import _init_distributed
    from passl_amd.hip import config as hip_config, nn as hnn
    dev = hip_config.set_device('gpu')
    rank, world = _init_distributed(dev)
    assert world == 2
    N, size = 8, 64
    oracle = S.SimSiamOracle(seed=0, zero_init_residual=False, dtype=torch.float64, **U.SOLVER)
    model, opt = U.build_product(torch.float32)
    assert all(m._sync for m in model.modules() if isinstance(m, hnn._BatchNormBase))       # converted by the factory
    U.load_oracle_state(model, oracle)
    param_sync(model)
    model.train()
    xs = []
    for r in range(2):
        gen = torch.Generator().manual_seed(777 + r)
        xs.append((torch.randn(N, 3, size, size, generator=gen), torch.randn(N, 3, size, size, generator=gen)))
    loss = model([xs[rank][0].cuda(), xs[rank][1].cuda()])
    opt.clear_grad()
    loss.backward()
    grad_sync([{'params': [p for p in model.parameters() if p.requires_grad]}])      # mean over ranks
    lall = loss.detach().clone()
    dist.all_reduce(lall)
    torch.cuda.synchronize()
    ps = dict(model.named_parameters())
    if rank == 0:
        x1 = torch.cat([xs[0][0], xs[1][0]]).double()
        x2 = torch.cat([xs[0][1], xs[1][1]]).double()
        ref = oracle.forward_backward(x1, x2)
        o32 = S.SimSiamOracle(seed=0, zero_init_residual=False, **U.SOLVER)
        r32 = o32.forward_backward(x1.float(), x2.float())
        got = float(lall) / 2
        assert abs(got - float(ref['loss'])) < max(3e-6, 6 * abs(float(r32['loss']) - float(ref['loss']))), (got, float(ref['loss']))
        worst = 0.0
        for n, g in ref['grads'].items():
            a, b = ps[n].grad.double().cpu().reshape(-1), g.reshape(-1)
            err = float((a - b).norm() / b.norm().clamp_min(1e-30))
            e32 = float((r32['grads'][n].double().reshape(-1) - b).norm() / b.norm().clamp_min(1e-30))
            assert err <= max(5e-4, 8 * e32), (n, err, e32)
            worst = max(worst, err)
        sd = model.state_dict()
        for k, v in oracle.st.items():          # running statistics of the GLOBAL batch (advanced twice)
            if S.is_stat(k):
                e = float(((sd[k].cpu().double() - v).abs() / v.abs().clamp_min(0.1)).max())
                assert e < 2e-3, (k, e)
    opt.step()
    torch.cuda.synchronize()
    for name, t in (('arena_q', model.arena_q.flat), ('arena_p', model.arena_p.flat)):
        ref_t = t.clone()
        dist.broadcast(ref_t, src=0)
        assert torch.equal(ref_t, t), '%s differs between ranks (max abs diff %.3e)' % (name, float((ref_t - t).abs().max()))
    dist.barrier()
    if rank == 0:
        print('DP-OK simsiam %.6f worst-grad-l2=%.3e' % (got, worst), flush=True)
    dist.destroy_process_group()


def lp_run():
    """Two data-parallel ranks of the SimSiam linear-probe recipe through the v2 Engine (param_sync at start-up,
    the classifier arena's gradient all-reduce, MomentumLARC on identical averaged gradients, the evaluation pass with
    gathered scores / labels): the run must equal the ONE-process restatement on the concatenated batches — updated
    classifier and the evaluation metric — and the replicas must end bit-identical."""
    from oracle import linprobe_v2 as L
    from passl_amd.engine.engine import Engine
    from passl_amd.utils.config import get_config
    classes, bs, size = 40, 8, 64
    cfg = get_config(os.path.join(ROOT, 'configs', 'v2', 'simsiam_resnet50_lp_synthetic.yaml'),
                     ['Global.epochs=1', 'Global.print_batch_step=1', 'Global.output_dir=%s' % os.environ['PASSL_DP_OUT'],
                      'Model.class_num=%d' % classes, 'DataLoader.Train.dataset.num_classes=%d' % classes,
                      'DataLoader.Eval.dataset.num_classes=%d' % classes,
                      'DataLoader.Train.dataset.num_samples=%d' % (2 * 2 * bs),
                      'DataLoader.Train.dataset.image_size=%d' % size,
                      'DataLoader.Train.sampler.batch_size=%d' % bs, 'DataLoader.Eval.dataset.num_samples=24',
                      'DataLoader.Eval.dataset.image_size=%d' % size, 'DataLoader.Eval.sampler.batch_size=%d' % bs])
    cfg.DataLoader.Train.dataset.num_batches_cached = 2
    cfg.DataLoader.Eval.dataset.num_batches_cached = 2
    eng = Engine(cfg, mode='train')
    rank, world = cfg['Global']['rank'], cfg['Global']['world_size']
    assert world == 2 and eng.grad_reducer is not None
    oracle = L.LinearProbeOracle('simsiam', class_num=classes, seed=0, optimizer='MomentumLARC', lr=1.6, momentum=0.9,
                                 weight_decay=0.0, trust_coefficient=0.001, clip=False)
    missing, unexpected = eng.model.load_state_dict({k: v.float() for k, v in oracle.st.items()}, strict=False)
    assert not missing and not unexpected
    eng.train()
    torch.cuda.synchronize()

    def batches(part, seed, n):
        ds = getattr(eng, part).inner.dataset
        out = []
        for r in range(world):
            gen = torch.Generator().manual_seed(seed + r)
            out.append([ds.make_batch(gen, bs) for _ in range(n)])
        return out
    if rank == 0:
        tr = batches('train_dataloader', 1234, 2)
        for s in range(2):
            oracle.train_step(torch.cat([tr[0][s][0], tr[1][s][0]]), torch.cat([tr[0][s][1], tr[1][s][1]]))
        for n, ref in (('fc.weight', oracle.st['fc.weight']), ('fc.bias', oracle.st['fc.bias'])):
            got = dict(eng.model.named_parameters())[n].detach().cpu().double()
            err = float((got - ref.double()).norm() / ref.double().norm())
            assert err < 2e-4, (n, err)
        ev = batches('eval_dataloader', 4321, 2)
        rows = []
        for r in range(world):
            rows.append((ev[r][0][0], ev[r][0][1]))
            rows.append((ev[r][1][0][:4], ev[r][1][1][:4]))        # 12 rows per rank: the last batch holds 4
        ref = oracle.evaluate(rows)
        got = eng.validate_loop.latest_model_metric
        assert abs(got['top1'] - ref['top1']) < 1e-6 and abs(got['top5'] - ref['top5']) < 1e-6, (got, ref)
    t = eng.model.arena_q.flat
    ref_t = t.clone()
    dist.broadcast(ref_t, src=0)
    assert torch.equal(ref_t, t), 'classifier differs between ranks'
    m = torch.tensor([eng.validate_loop.latest_model_metric['top1']], dtype=torch.float64, device=t.device)
    m0 = m.clone()
    dist.broadcast(m0, src=0)
    assert torch.equal(m, m0), 'evaluation metric differs between ranks'
    dist.barrier()
    if rank == 0:
        print('DP-OK lp %.6f' % eng.validate_loop.latest_model_metric['loss'], flush=True)
    dist.destroy_process_group()


def simsiam_engine_run():
    """Two data-parallel ranks of the SimSiam pre-training recipe through the v2 Engine: two trainable arenas (encoder /
    predictor parameter groups).  Default: one OVERLAPPED reducer per arena (core/sync_utils.py:ReducerGroup);
    PASSL_DP_BLOCKING_GROUPS=1: the loop's blocking grad_sync (the reference's passl/core/sync_utils.py:18-43).  Prints a
    digest of both arenas after three steps: the two modes must agree bit for bit (the caller compares), and the
    replicas must be identical."""
    import hashlib
    from passl_amd.core.sync_utils import ReducerGroup
    from passl_amd.engine.engine import Engine
    from passl_amd.utils.config import get_config
    cfg = get_config(os.path.join(ROOT, 'configs', 'v2', 'simsiam_resnet50_pt_synthetic.yaml'),
                     ['Global.epochs=1', 'Global.print_batch_step=1', 'Global.output_dir=%s' % os.environ['PASSL_DP_OUT'],
                      'DataLoader.Train.dataset.num_samples=%d' % (2 * 3 * 8), 'DataLoader.Train.dataset.image_size=64',
                      'DataLoader.Train.sampler.batch_size=8'])
    cfg.DataLoader.Train.dataset.num_batches_cached = 3
    eng = Engine(cfg, mode='train')
    world = cfg['Global']['world_size']
    blocking = os.environ.get('PASSL_DP_BLOCKING_GROUPS') == '1'
    assert world == 2
    assert (eng.grad_reducer is None) if blocking else isinstance(eng.grad_reducer, ReducerGroup)
    eng.train()
    torch.cuda.synchronize()
    arch = getattr(eng.model, 'arch', eng.model)
    h = hashlib.sha1()
    for a in arch.trainable_arenas():
        t = a.flat
        ref_t = t.clone()
        dist.broadcast(ref_t, src=0)
        assert torch.equal(ref_t, t), 'arena differs between ranks'
        h.update(t.detach().cpu().numpy().tobytes())
    dist.barrier()
    if cfg['Global']['rank'] == 0:
        print('DP-OK simsiam_engine digest=%s' % h.hexdigest(), flush=True)
    dist.destroy_process_group()


def main():
    workload = sys.argv[1]
    if workload == 'lp':
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        return lp_run()
    if workload == 'simsiam_engine':
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        return simsiam_engine_run()
    if workload == 'simsiam':
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        return simsiam_run()
    if workload == 'mocov3':
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        return mocov3_run()
    from passl_amd.engine.trainer import Trainer
    from passl_amd.hooks import OptimizerHook, LRSchedulerHook
    from passl_amd.utils.config import get_config
    path, ov = OVERRIDES[workload]
    # a non-zero seed: `seed: 0` means UNSEEDED in the reference's Trainer (trainer.py:105), and the
    # single-rank arms of test_rccl_world1 must start from the same weights in separate processes
    cfg = get_config(os.path.join(ROOT, path), ov + ['compute_dtype=fp32', 'seed=7'])
    if workload == 'clipx':
        cfg.model.multi_rank = True
    if workload == 'mae':
        cfg.model.architecture.img_size = 64
    if workload in ('moco', 'moco_shuffle'):
        cfg.model.K = 256
    if workload == 'moco_shuffle':
        cfg.model.shuffle_bn = True
    if workload == 'linprobe':
        cfg.model.head.num_classes = 16
        cfg.custom_config = []                        # no EvaluateHook in the 3-step run
    cfg.timestamp = ''
    tr = Trainer(cfg)
    forced = os.environ.get('PASSL_DP_FORCE') == '1'
    plain = 'WORLD_SIZE' not in os.environ          # reference arm: no process group, no collectives
    if plain:
        assert tr.world_size == 1 and tr.grad_reducer is None and not dist.is_initialized()
        return plain_run(tr, workload)
    assert tr.world_size == int(os.environ['WORLD_SIZE']) and tr.grad_reducer is not None
    assert tr.world_size > 1 or forced
    if os.environ.get('PASSL_EXPECT_BACKEND'):
        assert dist.get_backend() == os.environ['PASSL_EXPECT_BACKEND'], dist.get_backend()
    tr.mode = 'train'
    tr.model.train()
    opt_hook = next(h for h in tr.hooks if isinstance(h, OptimizerHook))
    lr_hook = next(h for h in tr.hooks if isinstance(h, LRSchedulerHook))
    data = next(iter(tr.train_dataloader))
    # ranks draw different batches (seed + rank)
    t0 = data[0].double().sum().reshape(1)
    both = [torch.zeros_like(t0) for _ in range(tr.world_size)]
    dist.all_gather(both, t0)
    assert len({float(b) for b in both}) == tr.world_size, 'ranks must see different data'

    def flat_params():
        return torch.cat([p.detach().reshape(-1).double() for p in tr.model.parameters()])

    def same_everywhere(t, what):
        ref = t.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, t), '%s differs between ranks (max abs diff %.3e)' % (
            what, float((ref - t).abs().max()))

    same_everywhere(flat_params(), 'initial parameters')
    p0 = flat_params()
    losses = three_steps(tr)
    loss = losses[-1]
    assert loss == loss and abs(loss) < 1e4, loss
    p1 = flat_params()
    assert float((p1 - p0).abs().max()) > 0, 'parameters did not move'
    # identical averaged gradients on every rank <=> replicas stay bit-identical
    same_everywhere(p1, 'parameters after 3 steps')
    if workload in ('moco', 'moco_shuffle'):
        same_everywhere(tr.model.queue.double(), 'queue (gathered keys)')
        assert tr.model._ptr == 3 * 8 * tr.world_size
    dist.barrier()
    if tr.rank == 0:
        # digest of the final parameters: the world-1 RCCL run must equal the collective-free run
        print('DP-OK %s %.6f digest=%.17g losses=%s' % (workload, loss, float(p1.sum()),
                                                       ','.join('%.9g' % l for l in losses)), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
