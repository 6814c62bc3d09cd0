from ...utils.registry import Registry, build_from_config

BACKBONES = Registry('BACKBONE')


def build_backbone(cfg):
    return build_from_config(cfg, BACKBONES)
