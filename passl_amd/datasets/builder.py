"""DATASETS registry + build_dataloader (reference passl_v110/datasets/builder.py:25-105).

Only the synthetic two-view source is built: the benchmark and the parity tests use synthetic
tensors, and the reference's PIL/cv2 augmentation pipeline is outside the accelerated path
(SURVEY §2.1 row 9).  ``name: ImageNet`` from the reference configs resolves to a class that
explains this instead of silently substituting data."""
from ..utils.registry import Registry, build_from_config

DATASETS = Registry('DATASET')


def build_dataset(cfg):
    return build_from_config(cfg, DATASETS)


def build_dataloader(cfg, device):
    """cfg = dataloader.train block: {loader, sampler, dataset}.  Returns (loader, mixup_fn)."""
    from .synthetic import SyntheticLoader
    ds_cfg = dict(cfg['dataset'])
    sampler = cfg.get('sampler', {})
    dataset = build_dataset(ds_cfg)
    loader = SyntheticLoader(dataset, batch_size=sampler.get('batch_size', 32), device=device,
                             drop_last=sampler.get('drop_last', True))
    ring = int((cfg.get('loader', None) or {}).get('host_ring', 0) or 0)
    if ring:
        from .synthetic import HostRingLoader
        loader = HostRingLoader(loader, ring=ring)       # batches move host -> device one step ahead of the step
    return loader, None
