"""bench.py — MoCo-v2 ResNet-50 two-view training step on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 5

One "step" = the full hot path of BASELINE.json configs[1] on one resident synthetic batch:
q forward (train BN), key-encoder EMA, k forward, fused InfoNCE, enqueue, backward, gradient
all-reduce (N>1), momentum-SGD, lr step — driven through the Trainer's hooks (OptimizerHook,
LRSchedulerHook).  Prints ONE JSON line (rank 0).  `value` = images/s over all ranks, where one
image = one two-view sample (PASSL's own `ips`, passl/engine/loops/loop.py:102-104).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402

FLOP_PER_SAMPLE = 32.77e9      # 2*[(3+1)*(4.0871+0.00446) + 2*0.00839] GMAC
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None,
                    help='per-GPU batch (default: 256 = BASELINE configs[1]; simclr: 64; clip: 128)')
    ap.add_argument('--workload', default='moco', choices=['moco', 'simclr', 'mae', 'clip', 'linprobe'],
                    help="moco = BASELINE.json's metric (default); simclr / mae / clip = the SimCLR, MAE and "
                         'CLIP rows (extra measurements: no-maxpool R50 + NT-Xent+CO2 + LARS; ViT-B/16 MAE; '
                         'CLIP ViT-B/32 image-text pairs)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true',
                    help='do not bracket the MFMA kernels with HIP events (use under rocprofv3)')
    return ap.parse_args()


def cpu_baseline():
    """The oracle (CPU restatement of the reference step) on the host cores, cfg-1 shape."""
    from oracle.moco import MoCoOracle
    n = 32
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    o = MoCoOracle(K=65536, seed=0)
    gen = torch.Generator().manual_seed(1234)
    xq = torch.randn(n, 3, 224, 224, generator=gen)
    xk = torch.randn(n, 3, 224, 224, generator=gen)
    o.train_step(xq, xk)                              # warm-up
    times = []
    for _ in range(2):
        t = time.perf_counter()
        o.train_step(xq, xk)
        times.append(time.perf_counter() - t)
    sec = sorted(times)[0]
    return {'value': round(n / sec, 3), 'unit': 'images/sec', 'cores': torch.get_num_threads(),
            'kind': 'port',
            'sample': 'oracle (torch-CPU fp32 restatement; Paddle is not installable) full step, '
                      'N=32 two-view 224^2, K=65536, best of 2 after 1 warm-up, %.2f s/step' % sec}


def pmc_traffic(args):
    """HBM bytes per igemm launch (a number) from the committed PMC passes of this same command
    (profiles/r01_pmc_traffic.json, made by tools/pmc_summary.py); null for any other workload/shape."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    if args.workload != 'moco' or args.batch != 256 or args.dtype != 'bf16' or not os.path.exists(path):
        return None
    with open(path) as f:
        z = json.load(f)
    return z['igemm_all_variants']['hbm_bytes_per_launch']


def main():
    args = parse()
    simclr, mae, clip = args.workload == 'simclr', args.workload == 'mae', args.workload == 'clip'
    linprobe = args.workload == 'linprobe'
    if args.batch is None:
        args.batch = 64 if simclr else (128 if clip else 256)
    # SimCLR: 2 views x (fwd + bwd = 3) x 15.99 GMAC x 2 FLOP per two-view sample;
    # MAE ViT-B/16: 3 x 9.78 GMAC x 2 FLOP per image
    # CLIP ViT-B/32: 4.41 GMAC (image, 50 tokens) + 2.98 GMAC (text, 77 tokens) per pair
    flop_per_sample = 2 * 3 * 15.99e9 * 2 if simclr else (3 * 9.78e9 * 2 if mae else FLOP_PER_SAMPLE)
    if clip:
        flop_per_sample = 3 * 7.39e9 * 2
    if linprobe:                        # frozen trunk forward only (4.087 GMAC) + the fc
        flop_per_sample = 4.09e9 * 2
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, 'launch with torchrun --nproc-per-node %d' % args.gpus
    assert torch.cuda.is_available(), 'bench.py runs the HIP path: an MI355X is required'

    from passl_amd.engine.trainer import Trainer
    from passl_amd.hip import lib as L, ops
    from passl_amd.hooks import OptimizerHook, LRSchedulerHook
    from passl_amd.utils.config import get_config

    cfg = get_config(os.path.join(ROOT, 'configs/simclr/simclr_r50_synthetic.yaml' if simclr else
                                  ('configs/mae/mae_vit_b_synthetic.yaml' if mae else
                                   ('configs/clip/vit-b-32_synthetic.yaml' if clip else
                                    ('configs/moco/moco_clas_r50_synthetic.yaml' if linprobe else
                                     'configs/moco/moco_v2_r50_synthetic.yaml')))),
                     ['dataloader.train.sampler.batch_size=%d' % args.batch,
                      'compute_dtype=%s' % args.dtype])
    cfg.timestamp = ''
    trainer = Trainer(cfg)
    trainer.mode = 'train'
    trainer.model.train()
    opt_hook = next(h for h in trainer.hooks if isinstance(h, OptimizerHook))
    lr_hook = next(h for h in trainer.hooks if isinstance(h, LRSchedulerHook))
    data = next(iter(trainer.train_dataloader))

    def step():
        trainer.current_iter += 1
        trainer.outputs = trainer.model(*data, total_iters=trainer.total_iters,
                                        current_iter=trainer.current_iter, mixup_fn=None)
        opt_hook.train_iter_end(trainer)
        lr_hook.train_iter_end(trainer)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    lib = L.load()
    timing = not args.no_kernel_timing
    flops = {'igemm': 0.0}
    if timing:
        lib.passl_hip_prof_enable(1)
        real_igemm = ops.conv_igemm

        def counting_igemm(d, *a, **k):
            kdim = 147 if (d.R, d.S, d.C) == (7, 1, 32) else d.R * d.S * d.C   # stem: real taps
            flops['igemm'] += 2.0 * d.N * d.OP * d.OQ * d.NCOLS * kdim
            return real_igemm(d, *a, **k)
        ops.conv_igemm = counting_igemm
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    loss = float(trainer.outputs['loss'].detach())

    kern = {}
    if timing:
        import ctypes
        ops.conv_igemm = real_igemm
        for cls, name in ((0, 'igemm'), (1, 'wgrad')):
            ms, n = ctypes.c_double(), ctypes.c_int64()
            lib.passl_hip_prof_collect(cls, ctypes.byref(ms), ctypes.byref(n))
            kern[name] = (ms.value, n.value)
        lib.passl_hip_prof_enable(0)

    t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ips = args.batch * world * args.steps / elapsed
        peak = PEAK_BF16_TFLOPS if args.dtype == 'bf16' else PEAK_F32_TFLOPS
        out = {
            'metric': ('images/sec/node, linear probe on frozen R50 bs%d/GPU' % args.batch) if linprobe else
            ('image-text pairs/sec/node, CLIP ViT-B/32 bs%d/GPU' % args.batch) if clip else
            ('images/sec/node, MAE ViT-B/16 mask 0.75 bs%d/GPU' % args.batch) if mae else
            'images/sec/node (2-view), %s bs%d/GPU' % (
                'SimCLR R50 (no stem max-pool)' if simclr else 'MoCo-v2 R50', args.batch),
            'value': round(ips, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1000 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': ('linear probe: frozen ResNet-50 (fused inference BN) + fc 2048->1000 %s, '
                                    'bs=%d/GPU, 224^2 synthetic labelled images, momentum-SGD on the fc '
                                    '(configs/moco/moco_clas_r50.yaml)' if linprobe else
                                    'CLIP ViT-B/32 + 12-layer causal text transformer %s, bs=%d/GPU, 224^2 '
                                    'synthetic images + 77-token synthetic captions, AdamW (CLIP row; '
                                    'configs/clip/vit-b-32.yaml)' if clip else
                                    'MAE ViT-B/16 %s, bs=%d/GPU, 224^2 synthetic images, mask 0.75 (50 '
                                    'encoder / 197 decoder tokens), norm_pix_loss, AdamW (MAE row; '
                                    'BASELINE configs[3])' if mae else
                                    'SimCLR ResNet-50 (no stem max-pool) %s, bs=%d/GPU, 2x224^2 '
                                    'synthetic views, NT-Xent+CO2 T=0.1, LARS (SimCLR row; '
                                    'BASELINE configs[2] shape at a smaller per-GPU batch)'
                                    if simclr else
                                    'MoCo-v2 ResNet-50 %s, bs=%d/GPU, 2x224^2 synthetic views, '
                                    'queue=65536, dim=128, T=0.2, m=0.999, momentum-SGD (BASELINE '
                                    'configs[1])') % (args.dtype, args.batch),
                       'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                       'views_per_sec': round(2 * ips, 2), 'final_loss': round(loss, 4)},
            'step_flop_roofline': {
                'algorithmic_gflop_per_sample': flop_per_sample / 1e9,
                'achieved_tflops_per_gpu': round(ips / world * flop_per_sample / 1e12, 2),
                'frac_of_peak': round(ips / world * flop_per_sample / 1e12 / peak, 5)},
        }
        if timing and kern.get('igemm', (0, 0))[1] > 0:
            ms, n = kern['igemm']
            ach = flops['igemm'] / (ms * 1e-3) / 1e12
            out['roofline'] = {
                'kernel': 'igemm_kernel (implicit-GEMM conv fwd + dgrad + linear)', 'bound': 'mfma',
                'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(ach / peak, 5), 'traffic': pmc_traffic(args),
                'traffic_unit': 'HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, '
                                'profiles/r01_pmc_traffic.json)',
                'launches': int(n), 'avg_launch_us': round(1000 * ms / n, 2),
                'algorithmic_gflop_per_launch': round(flops['igemm'] / n / 1e9, 3),
                'share_of_step_time': round(ms / (1000 * elapsed), 4)}
            wms, wn = kern.get('wgrad', (0, 0))
            if wn:
                out['roofline']['wgrad_kernel_ms_per_step'] = round(wms / args.steps, 3)
                out['roofline']['igemm_kernel_ms_per_step'] = round(ms / args.steps, 3)
        if world == 1 and not args.no_cpu_baseline and args.workload == 'moco':
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
