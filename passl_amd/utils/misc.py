"""Running-average meter used by the hook bus and the v2 loops.

Serves both spellings of the reference: passl_v110/utils/misc.py:17-39 passes format specs WITH the colon
(``AverageMeter('loss', ':.4e')``, printed through ``str()``), passl/utils/misc.py:33-76 passes them without
(``AverageMeter('batch_cost', '.5f', postfix=' s')``, printed through ``.mean / .value / .total``).
"""

__all__ = ['AverageMeter']


class AverageMeter(object):
    def __init__(self, name='', fmt='f', postfix='', need_avg=True):
        self.name = name
        self.fmt = fmt
        self.postfix = postfix
        self.need_avg = need_avg
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def _show(self, x):
        return format(x, self.fmt.lstrip(':'))

    def __str__(self):
        return '%s: %s (%s)' % (self.name, self._show(self.val), self._show(self.avg))

    @property
    def total(self):
        return '%s_sum: %s%s' % (self.name, self._show(self.sum), self.postfix)

    @property
    def total_minute(self):
        return '%s %s%s min' % (self.name, self._show(self.sum / 60), self.postfix)

    @property
    def mean(self):
        return '%s: %s%s' % (self.name, self._show(self.avg), self.postfix) if self.need_avg else ''

    @property
    def value(self):
        return '%s: %s%s' % (self.name, self._show(self.val), self.postfix)
