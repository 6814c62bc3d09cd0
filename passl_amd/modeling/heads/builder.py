from ...utils.registry import Registry, build_from_config

HEADS = Registry('HEAD')


def build_head(cfg):
    return build_from_config(cfg, HEADS)
