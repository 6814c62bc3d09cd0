"""The native step plan (passl_amd/hip/replay.py) through the Trainer: for every workload whose model opts in, the
Trainer replays the recorded launch list by default and the run equals the eager Trainer's bit for bit — loss
trajectory and every parameter — including a step-varying learning rate, MAE's fresh masking noise per step (a live
host call between two plan segments) and the three-stream MoCo schedule.  Reference loop: passl_v110/engine/
trainer.py:287-337 + hooks/optimizer_hook.py:25-50."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')

CASES = [('configs/moco/moco_v2_r50_synthetic.yaml', 16, 1), ('configs/clip/vit-b-32_synthetic.yaml', 8, 1),
         ('configs/mae/mae_vit_b_synthetic.yaml', 8, 2), ('configs/simclr/simclr_r50_synthetic.yaml', 8, 1),
         ('configs/clip/vit-b-16_synthetic.yaml', 8, 1)]          # cross-rank InfoNCE path (one rank: no collectives)


def _run(cfg_path, batch, plan, steps=8):
    from passl_amd.engine.trainer import Trainer
    from passl_amd.utils.config import get_config
    cfg = get_config(os.path.join(ROOT, cfg_path),
                     ['dataloader.train.sampler.batch_size=%d' % batch, 'compute_dtype=bf16', 'seed=3'])
    cfg.timestamp = ''
    cfg.step_plan = plan
    tr = Trainer(cfg)
    assert (tr.step_graph is not None) == plan
    tr.mode = 'train'
    tr.model.train()
    data = next(iter(tr.train_dataloader))
    tr.call_hook('run_begin')
    tr.call_hook('train_epoch_begin')
    losses = []
    for _ in range(steps):
        tr.inner_iter = tr.current_iter % tr.iters_per_epoch
        tr.current_iter += 1
        tr.call_hook('train_iter_begin')
        tr.train_step(data)
        tr.call_hook('train_iter_end')
        losses.append(tr.outputs['loss'].detach().reshape(()).float().clone())
    torch.cuda.synchronize()
    arena = getattr(tr.model, 'arena_q', None) or getattr(tr.model, 'arena')
    out = dict(losses=torch.stack(losses).cpu(), flat=arena.flat.clone().cpu(), sg=tr.step_graph)
    return out


@pytest.mark.parametrize('cfg_path,batch,segments', CASES)
def test_trainer_replays_the_step_plan_by_default(cfg_path, batch, segments):
    eager = _run(cfg_path, batch, False)
    torch.cuda.empty_cache()
    plan = _run(cfg_path, batch, True)
    sg = plan['sg']
    assert sg.failed is None and not sg.foreign, (sg.failed, sg.foreign)
    assert sg.captured and sg.replays == 4, sg.replays                    # 3 eager warm-up steps, 1 recording step
    assert sg.info['segments'] == segments and sg.info['kernels'] > 200, sg.info
    print(cfg_path, sg.info)
    assert torch.isfinite(eager['losses']).all()
    assert torch.equal(eager['losses'].view(torch.int32), plan['losses'].view(torch.int32)), \
        (eager['losses'], plan['losses'])
    assert torch.equal(eager['flat'].view(torch.int32), plan['flat'].view(torch.int32))
    assert float(eager['losses'][0]) != float(eager['losses'][-1])


def test_stem_tail_scheduling_changes_nothing_but_the_stem_filter_rounding():
    """The backward pass's tail (hip/nn.py: WgradLink, hip/config.py: stem_wgrad_parts / stem_wgrad_main_last /
    side_urgent_rows): handing every weight gradient to the side stream at once is pure scheduling — bit-identical
    parameters — and the stem's weight gradient in two pieces of the batch (one per stream) sums the same products in
    another grouping: only the stem filter (64 x 3 x 7 x 7 values) may differ, and by fp32 rounding."""
    from passl_amd.hip import config
    saved = dict(config._state)
    try:
        base = _run(CASES[0][0], 16, False, steps=1)['flat']
        config._state['side_urgent_rows'] = 1
        urgent = _run(CASES[0][0], 16, False, steps=1)['flat']
        config._state.update(saved)
        config._state['stem_wgrad_parts'] = 1
        whole = _run(CASES[0][0], 16, False, steps=1)['flat']
        config._state.update(saved)
        config._state['stem_wgrad_main_last'] = False
        side_only = _run(CASES[0][0], 16, False, steps=1)['flat']
    finally:
        config._state.update(saved)
    assert config.stem_wgrad_parts() == 2 and config.stem_wgrad_main_last()        # the defaults `base` ran with
    assert torch.equal(base.view(torch.int32), urgent.view(torch.int32))
    for other in (whole, side_only):
        diff = (base != other).nonzero().flatten()
        assert diff.numel() <= 64 * 3 * 7 * 7, diff.numel()
        assert torch.allclose(base, other, rtol=1e-5, atol=1e-7)
        if diff.numel():
            assert int(diff.max() - diff.min()) < 64 * 3 * 7 * 7 + 64          # one contiguous slot of the arena


def test_reset_gives_the_plan_memory_back_and_the_next_plan_continues_the_run():
    """StepPlan.reset() (another batch shape, bench.py's eager instrumentation after the timed loop): the recorded step's
    private allocator pool goes back to the allocator — the recording has to pair `_cuda_endAllocateToPool` with
    `_cuda_releasePool`, otherwise the pool outlives its MemPool object (SimCLR R50 at 512 / GPU: 184 GB that the next
    eager step could not have) — and the plan recorded afterwards continues the run bit for bit."""
    from passl_amd.engine.trainer import Trainer
    from passl_amd.utils.config import get_config

    def run(plan, reset_after=None, steps=12):
        cfg = get_config(os.path.join(ROOT, CASES[0][0]),
                         ['dataloader.train.sampler.batch_size=32', 'compute_dtype=bf16', 'seed=3'])
        cfg.timestamp = ''
        cfg.step_plan = plan
        tr = Trainer(cfg)
        tr.mode = 'train'
        tr.model.train()
        data = next(iter(tr.train_dataloader))
        tr.call_hook('run_begin')
        tr.call_hook('train_epoch_begin')
        losses, mem = [], None
        for it in range(steps):
            tr.inner_iter = tr.current_iter % tr.iters_per_epoch
            tr.current_iter += 1
            tr.call_hook('train_iter_begin')
            tr.train_step(data)
            tr.call_hook('train_iter_end')
            losses.append(tr.outputs['loss'].detach().reshape(()).float().clone())
            if reset_after is not None and it == reset_after:
                assert tr.step_graph.captured
                torch.cuda.synchronize()
                tr.outputs = None
                before = torch.cuda.memory_reserved()
                tr.step_graph.reset()
                torch.cuda.empty_cache()
                mem = (before, torch.cuda.memory_reserved())
                assert not tr.step_graph.captured
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), tr.model.arena_q.flat.clone().cpu(), tr.step_graph, mem

    torch.cuda.empty_cache()
    le, fe, _, _ = run(False)
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    lp, fp, sg, (before, after) = run(True, reset_after=6)
    step_bytes = torch.cuda.max_memory_allocated() - base
    print('reserved before / after reset: %.2f / %.2f GB; one step allocates %.2f GB at its peak'
          % (before / 2 ** 30, after / 2 ** 30, step_bytes / 2 ** 30))
    assert before - after > 0.5 * step_bytes, (before, after, step_bytes)
    assert sg.captured and sg.replays == 4, sg.replays          # 4-6 replay; 7-9 warm up again, 10 records, 11 replays
    assert torch.equal(le.view(torch.int32), lp.view(torch.int32)), (le, lp)
    assert torch.equal(fe.view(torch.int32), fp.view(torch.int32))


def test_plan_kill_switch(monkeypatch):
    from passl_amd.engine.trainer import Trainer
    from passl_amd.utils.config import get_config
    monkeypatch.setenv('PASSL_PLAN', '0')
    cfg = get_config(os.path.join(ROOT, CASES[0][0]), ['dataloader.train.sampler.batch_size=8', 'compute_dtype=bf16'])
    cfg.timestamp = ''
    assert Trainer(cfg).step_graph is None


def test_tail_batch_runs_eagerly_and_replays_continue():
    """A batch of another shape than the recorded step (the tail batch of a `drop_last: False` loader, reference
    configs/simclr/simclr_r50_IM.yaml:88-90) is run with eager launches instead of raising; replays continue for the
    recorded shape; and the run equals an all-eager Trainer fed the same batches bit for bit (the scratch buffers the
    recorded launches point into stay pinned although the eager step may grow the shared workspace)."""
    from passl_amd.engine.trainer import Trainer
    from passl_amd.utils.config import get_config

    def run(plan):
        cfg = get_config(os.path.join(ROOT, CASES[0][0]),
                         ['dataloader.train.sampler.batch_size=16', 'compute_dtype=bf16', 'seed=3'])
        cfg.timestamp = ''
        cfg.step_plan = plan
        tr = Trainer(cfg)
        tr.mode = 'train'
        tr.model.train()
        data = next(iter(tr.train_dataloader))
        short = [d[:8].contiguous() if torch.is_tensor(d) else d for d in data] if isinstance(data, (list, tuple)) \
            else {k: (v[:8].contiguous() if torch.is_tensor(v) else v) for k, v in data.items()}
        tr.call_hook('run_begin')
        tr.call_hook('train_epoch_begin')
        losses = []
        for it in range(9):
            tr.inner_iter = tr.current_iter % tr.iters_per_epoch
            tr.current_iter += 1
            tr.call_hook('train_iter_begin')
            tr.train_step(short if it == 6 else data)
            tr.call_hook('train_iter_end')
            losses.append(tr.outputs['loss'].detach().reshape(()).float().clone())
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), tr.model.arena_q.flat.clone().cpu(), tr.step_graph

    le, fe, _ = run(False)
    torch.cuda.empty_cache()
    lp, fp, sg = run(True)
    assert sg.captured and sg.eager_fallbacks == 1 and sg.replays == 4, (sg.eager_fallbacks, sg.replays)
    assert torch.equal(le.view(torch.int32), lp.view(torch.int32)), (le, lp)
    assert torch.equal(fe.view(torch.int32), fp.view(torch.int32))
