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
