"""Throughput of a v2 task yaml through passl.engine.Engine (the models that have no v110 Trainer config: MoCo-v3,
SimSiam): images/s over K timed steps of train_one_step on the resident synthetic batch.
    python scratch/bench_v2.py configs/v2/mocov3_vit_base_pt_synthetic.yaml [batch] [compute dtype] [steps]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl.engine.engine import Engine
from passl_amd.utils.config import get_config

path = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dtype = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
cfg = get_config(path, ['DataLoader.Train.sampler.batch_size=%d' % batch, 'Global.output_dir=/tmp/bench_v2',
                        'DataLoader.Train.dataset.num_samples=%d' % (batch * 1000)])
cfg['Global']['compute_dtype'] = dtype
eng = Engine(cfg, mode='train')
eng.model.train()
batch_data = next(iter(eng.train_dataloader))
loop = eng.train_loop
for _ in range(6):
    loop.global_step += 1
    loop.train_one_step(batch_data)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    loop.global_step += 1
    _o, ld = loop.train_one_step(batch_data)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(json.dumps({'config': os.path.basename(path), 'model': cfg['Model']['name'], 'batch': batch, 'dtype': dtype,
                  'ms_per_step': round(dt * 1e3, 3), 'images_per_sec': round(batch / dt, 1),
                  'loss': float(ld['loss']), 'hbm_allocated_gb': round(torch.cuda.max_memory_allocated() / 1e9, 1)}))
