"""Learning-rate schedulers with paddle.optimizer.lr semantics (the reference registers Paddle's
own classes, passl_v110/solver/lr_scheduler.py:20-28): the constructor performs the first
``step()`` (last_epoch 0 -> base lr), ``step()`` advances one unit, ``get_lr()``/``__call__``
return the current value.  CosineAnnealingDecay uses the closed form
``eta_min + (lr-eta_min)*(1+cos(pi*t/T_max))/2`` (Paddle's recursive form agrees to fp error)."""
import math

from .builder import LRSCHEDULERS


class LRScheduler(object):
    def __init__(self, learning_rate=0.1, last_epoch=-1, verbose=False):
        self.base_lr = float(learning_rate)
        self.last_lr = float(learning_rate)
        self.last_epoch = last_epoch
        self.verbose = verbose
        self.step()

    def __call__(self):
        return self.last_lr

    def get_lr(self):
        raise NotImplementedError

    def step(self, epoch=None):
        if epoch is None:
            self.last_epoch += 1
        else:
            self.last_epoch = epoch
        self.last_lr = self.get_lr()

    def state_dict(self):
        return {'last_epoch': self.last_epoch, 'last_lr': self.last_lr}

    def set_state_dict(self, sd):
        self.last_epoch = sd['last_epoch']
        self.last_lr = sd['last_lr']


@LRSCHEDULERS.register()
class CosineAnnealingDecay(LRScheduler):
    def __init__(self, learning_rate, T_max, eta_min=0, last_epoch=-1, verbose=False):
        self.T_max = T_max
        self.eta_min = float(eta_min)
        super().__init__(learning_rate, last_epoch, verbose)

    def get_lr(self):
        return self.eta_min + (self.base_lr - self.eta_min) * \
            (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2


@LRSCHEDULERS.register()
class MultiStepDecay(LRScheduler):
    def __init__(self, learning_rate, milestones, gamma=0.1, last_epoch=-1, verbose=False):
        self.milestones = list(milestones)
        self.gamma = gamma
        super().__init__(learning_rate, last_epoch, verbose)

    def get_lr(self):
        n = sum(1 for m in self.milestones if self.last_epoch >= m)
        return self.base_lr * (self.gamma ** n)


@LRSCHEDULERS.register()
class LinearWarmup(LRScheduler):
    def __init__(self, learning_rate, warmup_steps, start_lr, end_lr, last_epoch=-1, verbose=False):
        self.learning_rate = learning_rate      # float or LRScheduler
        self.warmup_steps = warmup_steps
        self.start_lr, self.end_lr = start_lr, end_lr
        super().__init__(start_lr, last_epoch, verbose)

    def get_lr(self):
        if self.last_epoch < self.warmup_steps:
            return (self.end_lr - self.start_lr) * float(self.last_epoch) / float(self.warmup_steps) \
                + self.start_lr
        if isinstance(self.learning_rate, LRScheduler):
            self.learning_rate.step(self.last_epoch - self.warmup_steps)
            return self.learning_rate()
        return self.learning_rate


@LRSCHEDULERS.register()
class Cosine(LRScheduler):
    """The cosine half of ``CosineWarmup`` — passl_v110/solver/lr_scheduler.py:29-62: counted in the wrapper's
    post-warm-up iterations, period ``T_max - warmup_steps``; the reference's branch for
    ``(last_epoch - 1 - T_max) % (2 T_max) == 0`` (a restart step taken from the previous value) is kept."""

    def __init__(self, learning_rate, T_max, warmup_steps, eta_min=0, last_epoch=1, verbose=False):
        self.T_max, self.warmup_steps, self.eta_min = T_max, warmup_steps, eta_min
        super().__init__(learning_rate, last_epoch=last_epoch, verbose=verbose)
        self.last_epoch = last_epoch

    def get_lr(self):
        if self.last_epoch == 0:
            return self.base_lr
        if (self.last_epoch - 1 - self.T_max) % (2 * self.T_max) == 0:
            return self.last_lr + (self.base_lr - self.eta_min) * (1 - math.cos(math.pi / self.T_max)) / 2
        return self.eta_min + 0.5 * (self.base_lr - self.eta_min) * (
            1 + math.cos(math.pi * self.last_epoch / (self.T_max - self.warmup_steps)))


@LRSCHEDULERS.register()
class CosineWarmup(LinearWarmup):
    """Linear warm-up ``start_lr -> end_lr`` over ``warmup_steps``, then ``Cosine`` — passl_v110/solver/
    lr_scheduler.py:68-102 (every argument in iterations; build_lr_scheduler passes the yaml through)."""

    def __init__(self, learning_rate, warmup_steps, start_lr, end_lr, T_max, eta_min=0, last_epoch=-1,
                 verbose=False):
        lr_sch = Cosine(learning_rate, T_max, warmup_steps, eta_min=eta_min, last_epoch=last_epoch, verbose=verbose)
        super().__init__(learning_rate=lr_sch, warmup_steps=warmup_steps, start_lr=start_lr, end_lr=end_lr,
                         last_epoch=last_epoch)
        self.update_specified = False


@LRSCHEDULERS.register()
class Cosinesimclr(LRScheduler):
    """passl_v110/solver/lr_scheduler.py:105-114."""

    def __init__(self, learning_rate, T_max, last_epoch=-1, verbose=False):
        self.T_max = T_max
        super().__init__(learning_rate, last_epoch, verbose)

    def get_lr(self):
        return self.base_lr * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2


@LRSCHEDULERS.register()
class simclrCosineWarmup(LinearWarmup):
    """Linear warm-up 0 -> lr over ``warmup_steps`` iterations, then Cosinesimclr counted from the
    end of the warm-up — passl_v110/solver/lr_scheduler.py:117-139 (extra yaml keys such as
    total_images / learning_rate_scaling arrive as **kwargs exactly like in the reference)."""

    def __init__(self, lr, warmup_steps, T_max, current_iter=0, last_epoch=-1, warmup_epoch=10,
                 **kwargs):
        lr_sch = Cosinesimclr(lr, T_max, last_epoch=-1)
        super().__init__(learning_rate=lr_sch, warmup_steps=warmup_steps, start_lr=0.0, end_lr=lr,
                         last_epoch=last_epoch)
        self.update_specified = False


@LRSCHEDULERS.register()
class TimmCosine(LRScheduler):
    """passl/scheduler/lr_scheduler.py:22-77 (the v2 schedulers; MoCo-v3 / MAE-v2 configs): linear warm-up from
    ``warmup_start_lr`` over ``warmup_epoch`` epochs, then one cosine period down to ``eta_min``; with
    ``warmup_prefix`` the cosine is counted from the end of the warm-up over the remaining ``T_max - warmup``
    units.  ``decay_unit: step`` counts optimizer steps (``epochs * step_each_epoch`` in total).

    Two reference behaviours that differ from paddle.optimizer.lr's own classes, both kept:
      * the constructor never calls the base class' __init__, so it does NOT take the first ``step()``:
        ``last_epoch`` stays at the yaml's value (-1 by default);
      * the v2 optimizers do not read the cached ``last_lr``: passl/optimizer/optimizer.py:117-120 (``_get_lr``)
        evaluates ``get_lr()`` at the CURRENT ``last_epoch`` on every step.  ``__call__`` does the same here.
    With the loop's ``lr_step(global_step)`` after every optimizer step (contrastive_learning_loop.py:86-87,
    classification_loop.py:97-98) optimizer step n >= 2 runs at ``get_lr(n - 1)`` and step 1 at ``get_lr(-1)``:
    ``warmup_start_lr`` when there is a warm-up (MoCo-v3: the first step moves nothing).  Without a warm-up
    ``get_lr(-1)`` takes the warm-up branch (-1 < 0) and divides by ``warmup_steps = 0`` — in the reference too —
    which is why the recipes without one (SimSiam, the linear probes) set ``last_epoch: 0``."""

    def __init__(self, learning_rate, step_each_epoch, epochs, decay_unit='epoch', eta_min=0.0, warmup_epoch=0,
                 warmup_start_lr=0.0, warmup_prefix=False, verbose=False, last_epoch=-1, **kwargs):
        if warmup_epoch >= epochs:
            warmup_epoch = epochs
        if not isinstance(learning_rate, (float, int)):
            raise TypeError('The type of learning rate must be float, but received {}'.format(type(learning_rate)))
        assert decay_unit in ['step', 'epoch']
        self.learning_rate = learning_rate
        self.decay_unit = decay_unit
        if decay_unit == 'step':
            self.T_max = epochs * step_each_epoch
            self.warmup_steps = int(round(warmup_epoch * step_each_epoch))
        else:
            self.T_max = epochs
            self.warmup_steps = warmup_epoch
        self.eta_min = eta_min
        self.warmup_start_lr = warmup_start_lr
        self.warmup_prefix = warmup_prefix
        self.base_lr = float(learning_rate)
        self.last_lr = float(learning_rate)
        self.last_epoch = last_epoch
        self.verbose = verbose

    def get_lr(self):
        if self.last_epoch < self.warmup_steps:
            return float(max(0, self.last_epoch)) * (self.learning_rate - self.warmup_start_lr) / \
                float(self.warmup_steps) + self.warmup_start_lr
        last_epoch, T_max = self.last_epoch, self.T_max
        if self.warmup_prefix:
            last_epoch = last_epoch - self.warmup_steps
            T_max = self.T_max - self.warmup_steps
        cur_steps = last_epoch - (self.T_max * (last_epoch // self.T_max))
        return self.eta_min + 0.5 * (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * cur_steps / T_max))

    def __call__(self):
        return self.get_lr()
