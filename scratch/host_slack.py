"""How far ahead of the GPU does the host run?  Times the enqueue of K steps (no sync) against the
GPU's completion of the same steps."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.engine.trainer import Trainer
from passl_amd.utils.config import get_config
import logging
logging.getLogger('passl').setLevel(logging.WARNING)
wl = sys.argv[1] if len(sys.argv) > 1 else 'configs/moco/moco_v2_r50_synthetic.yaml'
bs = sys.argv[2] if len(sys.argv) > 2 else '256'
cfg = get_config(wl, ['dataloader.train.sampler.batch_size=%s' % bs, 'compute_dtype=bf16'])
cfg.timestamp = ''
tr = Trainer(cfg); tr.mode = 'train'; tr.model.train()
data = next(iter(tr.train_dataloader))
tr.call_hook('run_begin'); tr.call_hook('train_epoch_begin')
def step():
    tr.inner_iter = tr.current_iter % tr.iters_per_epoch
    tr.current_iter += 1
    tr.call_hook('train_iter_begin')
    tr.outputs = tr.model(*data, total_iters=tr.total_iters, current_iter=tr.current_iter, mixup_fn=tr.mixup_fn)
    tr.call_hook('train_iter_end')
for _ in range(6): step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('%s bs %s: host enqueue %.2f ms/step, GPU complete %.2f ms/step (host is %.0f%% of the step)' % (
    os.path.basename(wl), bs, 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K, 100 * (t1 - t0) / (t2 - t0)))
