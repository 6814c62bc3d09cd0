from ...utils.registry import Registry, build_from_config

MODELS = Registry('MODEL')


def build_model(cfg):
    return build_from_config(cfg, MODELS)
