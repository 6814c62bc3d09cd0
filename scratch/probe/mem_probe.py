"""What is live on the device between two steps?  (round 6: SimCLR R50 at 512 / GPU ran out of memory while the step
plan recorded: 184 GB in the plan's pool + 71 GB live in the general pool.)  Usage: python scratch/probe/mem_probe.py [batch]"""
import gc, os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from passl_amd.utils.config import get_config
from passl_amd.engine.trainer import Trainer

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = get_config(os.path.join(ROOT, 'configs/simclr/simclr_r50_synthetic.yaml'),
                 ['dataloader.train.sampler.batch_size=%d' % batch, 'compute_dtype=bf16'])
cfg.timestamp = ''
cfg.step_plan = False
tr = Trainer(cfg); tr.mode = 'train'; tr.model.train()
data = next(iter(tr.train_dataloader))
tr.call_hook('run_begin'); tr.call_hook('train_epoch_begin')
GB = 1 / 2 ** 30


def live(tag):
    torch.cuda.synchronize(); gc.collect()
    print('%-40s allocated %.2f GB  reserved %.2f GB  peak %.2f GB' % (
        tag, torch.cuda.memory_allocated() * GB, torch.cuda.memory_reserved() * GB, torch.cuda.max_memory_allocated() * GB))


def census(top=12):
    seen, by = set(), collections.Counter()
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                st = o.untyped_storage()
                if st.data_ptr() in seen:
                    continue
                seen.add(st.data_ptr())
                by[(tuple(o.shape), str(o.dtype))] += st.nbytes()
        except Exception:
            pass
    tot = sum(by.values())
    print('  python-visible device tensors: %.2f GB' % (tot * GB))
    for (shape, dt), n in by.most_common(top):
        print('   %8.3f GB  %s %s' % (n * GB, shape, dt))


def step():
    tr.inner_iter = tr.current_iter % tr.iters_per_epoch
    tr.current_iter += 1
    tr.call_hook('train_iter_begin'); tr.train_step(data); tr.call_hook('train_iter_end')


live('built')
for i in range(2):
    torch.cuda.reset_peak_memory_stats()
    step()
    live('after eager step %d (outputs held)' % i)
census()
tr.outputs = None
live('outputs dropped')
census()
torch.cuda.empty_cache()
live('after empty_cache')
