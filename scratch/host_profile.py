"""cProfile of the host side of K steps (where does the Python time of a launch-bound step go?)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.engine.trainer import Trainer
from passl_amd.utils.config import get_config
import logging
logging.getLogger('passl').setLevel(logging.WARNING)
wl = sys.argv[1] if len(sys.argv) > 1 else 'configs/clip/vit-b-32_synthetic.yaml'
bs = sys.argv[2] if len(sys.argv) > 2 else '128'
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
pr = cProfile.Profile()
pr.enable()
for _ in range(10): step()
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:6000])
