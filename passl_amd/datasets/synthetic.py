"""Synthetic two-view image source: x_q, x_k ~ N(0,1) fp32 [N,3,S,S] drawn from
``torch.Generator(seed = 1234 + rank)`` — post-NormalizeImage images are ~zero-mean/unit-variance
(configs/moco/moco_v2_r50.yaml:57-60).  Batches are generated on the host once and kept resident
in HBM, so the timed step never includes host->device copies."""
import os

import torch

from .builder import DATASETS


@DATASETS.register()
class SyntheticTwoView(object):
    def __init__(self, num_samples=1281167, image_size=224, seed=1234, num_batches_cached=1,
                 **ignored):
        self.num_samples = int(num_samples)
        self.image_size = int(image_size)
        self.seed = int(seed)
        self.num_batches_cached = int(num_batches_cached)

    def __len__(self):
        return self.num_samples


@DATASETS.register()
class SyntheticImageText(object):
    """(image, text) pairs shaped like the reference's TextImageDataset output
    (passl_v110/datasets/text_image_dataset.py): image ~ N(0,1) fp32 [3,S,S]; text int64
    [context_length] = random ids, the EOT (largest id = vocab_size-1) at a random position, zero
    padding after it."""

    def __init__(self, num_samples=75750, image_size=224, context_length=77, vocab_size=49408, seed=1234,
                 num_batches_cached=1, **ignored):
        self.num_samples, self.image_size = int(num_samples), int(image_size)
        self.context_length, self.vocab_size = int(context_length), int(vocab_size)
        self.seed, self.num_batches_cached = int(seed), int(num_batches_cached)

    def __len__(self):
        return self.num_samples

    def make_batch(self, gen, batch_size):
        s, T, V = self.image_size, self.context_length, self.vocab_size
        image = torch.randn(batch_size, 3, s, s, generator=gen)
        n = torch.randint(3, T + 1, (batch_size,), generator=gen)
        text = torch.randint(1, V - 2, (batch_size, T), generator=gen)
        pos = torch.arange(T).unsqueeze(0)
        text = torch.where(pos < (n - 1).unsqueeze(1), text, torch.zeros_like(text))
        text[torch.arange(batch_size), n - 1] = V - 1
        return image, text


@DATASETS.register()
class SyntheticLabeled(object):
    """(image, label) pairs shaped like the reference's ImageNet dataset with ``return_label: True``
    (passl_v110/datasets/imagenet.py): image ~ N(0,1) fp32 [3,S,S] plus a class-dependent offset (so a
    linear probe has something to fit), label int64; ``evaluate`` = imagenet.py:75-80."""

    def __init__(self, num_samples=1281167, image_size=224, num_classes=1000, seed=1234,
                 num_batches_cached=1, **ignored):
        self.num_samples, self.image_size = int(num_samples), int(image_size)
        self.num_classes, self.seed = int(num_classes), int(seed)
        self.num_batches_cached = int(num_batches_cached)

    def __len__(self):
        return self.num_samples

    def make_batch(self, gen, batch_size):
        s = self.image_size
        label = torch.randint(0, self.num_classes, (batch_size,), generator=gen)
        image = torch.randn(batch_size, 3, s, s, generator=gen)
        # a per-class colour cast: channel c shifted by cos(label * (c + 1))
        shift = torch.cos(label.double().unsqueeze(1) * torch.arange(1, 4).double()).float()
        return image + shift.view(batch_size, 3, 1, 1), label

    def evaluate(self, preds, labels, topk=(1, 5)):
        from ..modeling.heads.clas_head import accuracy
        eval_res = {}
        eval_res['acc1'], eval_res['acc5'] = accuracy(preds, labels, topk)
        return eval_res


@DATASETS.register()
class ImageNet(object):
    def __init__(self, **kwargs):
        raise NotImplementedError(
            'The ImageNet folder dataset + CPU augmentation pipeline of the reference is outside '
            'the MI355X hot path (SURVEY §2.1 row 9).  Use `-o dataloader.train.dataset.name='
            'SyntheticTwoView` for benchmarking/parity runs.')


@DATASETS.register()
class CIFAR10(object):
    """configs/simclr/simclr_r18_cifar10.yaml:26 (reference passl_v110/datasets/cifar.py:26-68)."""

    def __init__(self, **kwargs):
        raise NotImplementedError(
            'The CIFAR-10 download + PIL augmentation pipeline of the reference is outside the MI355X hot path '
            '(SURVEY §2.1 row 9).  Use `-o dataloader.train.dataset.name=SyntheticCIFAR10` (32 x 32 two-view synthetic '
            'batches; the remaining dataset keys of the yaml are ignored).')


@DATASETS.register()
class SyntheticCIFAR10(SyntheticTwoView):
    """Two N(0,1) views of CIFAR's shape (50 000 samples, 32 x 32): the stand-in that lets
    configs/simclr/simclr_r18_cifar10.yaml run with only the dataset NAME overridden."""

    def __init__(self, num_samples=50000, image_size=32, seed=1234, num_batches_cached=1, **ignored):
        super().__init__(num_samples=num_samples, image_size=image_size, seed=seed,
                         num_batches_cached=num_batches_cached)


@DATASETS.register()
class ImageFolder(object):
    """The v2 configs' dataset (reference passl/data/dataset/imagefolder_dataset.py)."""

    def __init__(self, **kwargs):
        raise NotImplementedError(
            'The image-folder dataset + CPU augmentation pipeline of the reference is outside the MI355X hot path '
            '(SURVEY §2.1 row 9).  Use `-o DataLoader.Train.dataset.name=SyntheticTwoView` for benchmarking / '
            'parity runs (the remaining dataset keys of the yaml are ignored by the synthetic source).')


class SyntheticLoader(object):
    def __init__(self, dataset, batch_size, device, drop_last=True):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.device = device
        world = int(os.environ.get('WORLD_SIZE', 1))
        rank = int(os.environ.get('RANK', 0))
        per_rank = len(dataset) // world
        self._len = per_rank // self.batch_size if drop_last else -(-per_rank // self.batch_size)
        # drop_last False: the last batch holds the remaining rows only (evaluation loaders of the v2 recipes)
        self._tail = 0 if drop_last else per_rank % self.batch_size
        gen = torch.Generator().manual_seed(dataset.seed + rank)
        s = dataset.image_size
        self._cache = []
        for _ in range(max(1, dataset.num_batches_cached)):
            if hasattr(dataset, 'make_batch'):
                self._cache.append(tuple(t.to(device) for t in dataset.make_batch(gen, self.batch_size)))
                continue
            xq = torch.randn(self.batch_size, 3, s, s, generator=gen)
            xk = torch.randn(self.batch_size, 3, s, s, generator=gen)
            self._cache.append((xq.to(device), xk.to(device)))

    def __len__(self):
        return max(self._len, 1)

    def __iter__(self):
        n = len(self)
        for i in range(n):
            b = self._cache[i % len(self._cache)]
            if self._tail and i == n - 1 and self._len > 0:
                b = tuple(t[:self._tail] for t in b)
            yield b


_RING_DIAG = os.environ.get('PASSL_RING_DIAG', '')


class HostRingLoader(object):
    """A loader whose batches MOVE: a ring of ``ring`` (>= 3) distinct pinned host batches; batch i's host -> device
    copy is issued one step ahead on a copy stream of its own into one of three device slots, so that the step that
    consumes it finds it resident and the compute stream never carries a PCIe transfer (the reference keeps the
    reader ahead of the step with DataLoader workers + ``use_shared_memory``, passl_v110/datasets/builder.py:60-105,
    and hands `data` to the model at trainer.py:318-321).

    Ordering: the copy into a slot waits (on the GPU) for an event recorded on the compute stream when the step that
    last read the slot had been enqueued; ``next()`` makes the compute stream wait for the copy's event.  With three
    slots and a look-ahead of one the slot being overwritten was read two steps ago.
    ``dataloader.train.loader.host_ring: n`` in a config selects it (build_dataloader)."""

    SLOTS = 3

    def __init__(self, inner, ring=3):
        assert ring >= 1
        self.inner = inner
        self.dataset = inner.dataset
        self.batch_size = inner.batch_size
        self.device = inner.device
        self._len = len(inner)
        rank = int(os.environ.get('RANK', 0))
        gen = torch.Generator().manual_seed(inner.dataset.seed + rank)
        s = inner.dataset.image_size
        self._host = []
        for _ in range(int(ring)):
            if hasattr(inner.dataset, 'make_batch'):
                b = tuple(inner.dataset.make_batch(gen, self.batch_size))
            else:
                b = (torch.randn(self.batch_size, 3, s, s, generator=gen),
                     torch.randn(self.batch_size, 3, s, s, generator=gen))
            self._host.append(tuple(t.pin_memory() if self.device.type == 'cuda' else t for t in b))
        self._cuda = self.device.type == 'cuda'
        self._slots = [tuple(torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in self._host[0])
                       for _ in range(self.SLOTS)] if self._cuda else None
        self._copy = torch.cuda.Stream(device=self.device) if self._cuda else None
        self._ready = [None] * self.SLOTS        # event of the copy stream: the slot holds its batch
        self._freed = [None] * self.SLOTS        # event of the compute stream: the slot's last reader is enqueued
        self._issued = 0                         # batches whose copy has been issued
        self._taken = 0                          # batches handed out
        self.bytes_per_batch = sum(t.numel() * t.element_size() for t in self._host[0])

    def __len__(self):
        return self._len

    def _issue(self):
        i = self._issued
        slot = i % self.SLOTS
        if self._freed[slot] is not None:
            self._copy.wait_event(self._freed[slot])
        with torch.cuda.stream(self._copy):
            if not (_RING_DIAG == 'noh2d' and i >= self.SLOTS):       # (diagnostic: every slot filled once, then no transfers)
                for d, h in zip(self._slots[slot], self._host[i % len(self._host)]):
                    d.copy_(h, non_blocking=True)
            self._ready[slot] = self._copy.record_event()
        self._issued += 1

    def take(self):
        """The next batch (device tensors of a slot); the following batch's transfer is issued before returning."""
        if not self._cuda:
            b = self._host[self._taken % len(self._host)]
            self._taken += 1
            return b
        cur = torch.cuda.current_stream(self.device)
        if self._taken > 0:
            # everything enqueued so far has read the slot handed out LAST time (the step that consumed it)
            self._freed[(self._taken - 1) % self.SLOTS] = cur.record_event()
        while self._issued < self._taken + 2:
            self._issue()                        # this batch (first call) and the one after it
        slot = self._taken % self.SLOTS
        cur.wait_event(self._ready[slot])
        self._taken += 1
        return self._slots[slot]

    def __iter__(self):
        for _ in range(len(self)):
            yield self.take()
