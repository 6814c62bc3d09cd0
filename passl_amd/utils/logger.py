"""Process logger: INFO to stdout on rank 0 (+ optional file), WARNING elsewhere
(reference: passl_v110/utils/logger.py)."""
import logging
import os
import sys

logger_initialized = []


def _rank():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except Exception:
        pass
    return int(os.environ.get('RANK', 0))


def setup_logger(output=None, name='passl'):
    logger = logging.getLogger(name)
    if name in logger_initialized:
        return logger
    logger.setLevel(logging.INFO)
    logger.propagate = False
    fmt = logging.Formatter('[%(asctime)s] %(name)s %(levelname)s: %(message)s', datefmt='%m/%d %H:%M:%S')
    rank = _rank()
    if rank == 0:
        ch = logging.StreamHandler(stream=sys.stdout)
        ch.setLevel(logging.DEBUG)
        ch.setFormatter(fmt)
        logger.addHandler(ch)
    else:
        logger.setLevel(logging.WARNING)
    if output is not None:
        filename = output if output.endswith(('.txt', '.log')) else os.path.join(output, 'log.txt')
        if rank > 0:
            filename = filename + '.rank{}'.format(rank)
        os.makedirs(os.path.dirname(filename) or '.', exist_ok=True)
        fh = logging.FileHandler(filename, mode='a')
        fh.setLevel(logging.DEBUG)
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    logger_initialized.append(name)
    return logger


def get_logger(name='passl', output=None):
    logger = logging.getLogger(name)
    if name in logger_initialized:
        return logger
    return setup_logger(name=name, output=output)
