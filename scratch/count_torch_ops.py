"""Which ATen ops (= torch-launched kernels / copies, not libpassl_hip launches) run inside one MoCo
training step?  TorchDispatchMode counts every dispatched op during a step, grouped by op and by the
innermost passl_amd / bench frame that issued it."""
import collections
import os
import sys
import traceback

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from passl_amd.engine.trainer import Trainer
from passl_amd.utils.config import get_config

BATCH = int(os.environ.get('BATCH', 64))
cfg = get_config(os.path.join(ROOT, 'configs/moco/moco_v2_r50_synthetic.yaml'),
                 ['dataloader.train.sampler.batch_size=%d' % BATCH, 'compute_dtype=bf16'])
cfg.timestamp = ''
tr = Trainer(cfg)
tr.mode = 'train'
tr.model.train()
data = next(iter(tr.train_dataloader))
tr.call_hook('run_begin'); tr.call_hook('train_epoch_begin')


def step():
    tr.inner_iter = tr.current_iter % tr.iters_per_epoch
    tr.current_iter += 1
    tr.call_hook('train_iter_begin')
    tr.outputs = tr.model(*data, total_iters=tr.total_iters, current_iter=tr.current_iter, mixup_fn=None)
    tr.call_hook('train_iter_end')


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = collections.Counter()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        self.ops[name] += 1
        site = '?'
        for fr in reversed(traceback.extract_stack()[:-1]):
            if 'passl_amd' in fr.filename or fr.filename.endswith('count_torch_ops.py'):
                site = '%s:%d' % (os.path.relpath(fr.filename, ROOT), fr.lineno)
                break
        self.sites[(name, site)] += 1
        return func(*args, **(kwargs or {}))


for _ in range(3):
    step()
torch.cuda.synchronize()
with Counter() as c:
    step()
torch.cuda.synchronize()
skip = ('aten.view', 'aten.detach', 'aten.empty', 'aten.alias', 'aten._unsafe_view', 'aten.slice', 'aten.select',
        'aten.as_strided', 'aten.permute', 'aten.t.', 'aten.expand', 'aten.reshape', 'aten.transpose',
        'aten.unsqueeze', 'aten.squeeze', 'aten.is_', 'aten.empty_like', 'aten.new_empty')
print('# ATen ops dispatched in ONE MoCo step (views / allocations without a kernel omitted)')
for (name, site), n in sorted(c.sites.items(), key=lambda kv: -kv[1]):
    if name.startswith(skip):
        continue
    print('%5d  %-40s %s' % (n, name, site))
