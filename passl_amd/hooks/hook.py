"""Hook interface of the Trainer (the event names and the train_/val_ -> generic forwarding are the
contract of passl_v110/hooks/hook.py:16-69; hooks in this package and user hooks written for the
reference override the same methods).

The event table is data: ``STAGES`` lists the generic events, every ``train_<event>`` / ``val_<event>``
forwards to ``<event>`` unless a subclass overrides it."""

STAGES = ('epoch_begin', 'epoch_end', 'iter_begin', 'iter_end')


def _forward_to(event):
    def handler(self, trainer):
        return getattr(self, event)(trainer)
    handler.__name__ = event
    handler.__doc__ = 'forwards to %s()' % event
    return handler


class Hook:
    """Base class: every event is a no-op."""

    def run_begin(self, trainer):
        """once, before the first epoch"""

    def run_end(self, trainer):
        """once, after the last iteration"""

    # ---- periodic predicates used by the hooks (1-based counters, n <= 0 never fires)
    @staticmethod
    def _hits(count, n):
        return n > 0 and count % n == 0

    def every_n_epochs(self, trainer, n):
        return self._hits(trainer.current_epoch + 1, n)

    def every_n_iters(self, trainer, n):
        return self._hits(trainer.current_iter + 1, n)

    def every_n_inner_iters(self, trainer, n):
        return self._hits(trainer.inner_iter, n)

    def end_of_epoch(self, trainer):
        return trainer.iters_per_epoch == trainer.inner_iter + 1


for _event in STAGES:
    setattr(Hook, _event, (lambda self, trainer: None))
    for _phase in ('train', 'val'):
        setattr(Hook, '%s_%s' % (_phase, _event), _forward_to(_event))
del _event, _phase
