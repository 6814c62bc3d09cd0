"""Native step plan (passl_amd/hip/replay.py) per workload: which ATen launches are still inside the step (the plan
refuses to replay a step that has any), what the recorded plan looks like, and eager vs replayed step time + host time.
    python scratch/plan_probe.py moco clip clip16 mae simclr [--steps 20]"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch

import bench as B
from passl_amd.engine.trainer import Trainer
from passl_amd.utils.config import get_config
import logging
logging.getLogger('passl').setLevel(logging.WARNING)

names = [a for a in sys.argv[1:] if not a.startswith('--')] or ['moco']
steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 20


def run(name, plan, force_safe):
    cfg_path, batch = B.WORKLOADS[name][0], B.WORKLOADS[name][1]
    cfg = get_config(os.path.join(ROOT, cfg_path), ['dataloader.train.sampler.batch_size=%d' % batch,
                                                    'compute_dtype=bf16'])
    cfg.timestamp = ''
    cfg.step_plan = plan
    if force_safe:
        os.environ['PASSL_PLAN_STRICT'] = '0'
    tr = Trainer.__new__(Trainer)
    # graph_safe is read in Trainer.__init__ (_build_step_graph): force it on the class of the model being built
    import passl_amd.engine.trainer as T
    orig = T.build_model

    def patched(c):
        m = orig(c)
        if force_safe:
            type(m).graph_safe = True
        return m
    T.build_model, saved = patched, T.build_model
    try:
        tr.__init__(cfg)
    finally:
        T.build_model = saved
    tr.mode = 'train'
    tr.model.train()
    data = next(iter(tr.train_dataloader))
    tr.call_hook('run_begin')
    tr.call_hook('train_epoch_begin')

    def step():
        tr.inner_iter = tr.current_iter % tr.iters_per_epoch
        tr.current_iter += 1
        tr.call_hook('train_iter_begin')
        tr.train_step(data)
        tr.call_hook('train_iter_end')
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    host_idle = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        host_idle.append(1e3 * (time.perf_counter() - t1))
    torch.cuda.synchronize()
    sg = tr.step_graph
    rec = dict(host_ms_into_idle_gpu=[round(h, 3) for h in host_idle], workload=name, plan=plan, ms_per_step=round(1e3 * el / steps, 3), host_ms_per_step=round(1e3 * host / steps, 3),
               per_sec=round(batch * steps / el, 1), loss=float(tr.outputs['loss'].detach()),
               captured=bool(sg is not None and sg.captured), info=getattr(sg, 'info', None),
               failed=getattr(sg, 'failed', None),
               foreign=['%s x%d @ %s' % (n, c, s) for (n, s), c in sorted(getattr(sg, 'foreign', {}).items())])
    print(json.dumps(rec), flush=True)
    del tr
    torch.cuda.empty_cache()


for n in names:
    for plan in (False, True):
        try:
            run(n, plan, force_safe=plan and '--force' in sys.argv)
        except Exception as e:                       # keep going: this is a survey
            import traceback
            traceback.print_exc()
            print(json.dumps(dict(workload=n, plan=plan, error=repr(e)[:300])), flush=True)
