from ...utils.registry import Registry, build_from_config

NECKS = Registry('NECK')


def build_neck(cfg):
    return build_from_config(cfg, NECKS)
