"""Run setup: output dir, timestamp, logger (reference: passl_v110/utils/setup.py:22-53)."""
import os
import time

from .logger import setup_logger


def setup(args, cfg):
    if getattr(args, 'evaluate_only', False):
        cfg.is_train = False
    else:
        cfg.is_train = True
    cfg.timestamp = time.strftime('-%Y-%m-%d-%H-%M', time.localtime())
    cfg.output_dir = os.path.join(cfg.get('output_dir', 'output_dir'),
                                  os.path.splitext(os.path.basename(str(args.config_file)))[0] + cfg.timestamp)
    os.makedirs(cfg.output_dir, exist_ok=True)
    logger = setup_logger(cfg.output_dir)
    logger.info('Configs: {}'.format(cfg))
    if getattr(args, 'device', None):
        cfg.device = args.device
    return logger
