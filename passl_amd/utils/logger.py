"""Process logger (role of passl_v110/utils/logger.py): rank 0 logs INFO to stdout, other ranks only
WARNING and above; with ``output`` every rank also appends to a file (``<output>/log.txt`` or the given
.txt/.log path, suffixed ``.rank<r>`` off rank 0).  Handlers are installed once per logger name."""
import logging
import os
import sys

logger_initialized = []
_FORMAT = logging.Formatter('[%(asctime)s] %(name)s %(levelname)s: %(message)s', datefmt='%m/%d %H:%M:%S')


def _rank():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except Exception:
        pass
    return int(os.environ.get('RANK', 0))


def _log_path(output, rank):
    path = output if output.endswith(('.txt', '.log')) else os.path.join(output, 'log.txt')
    return path if rank == 0 else '%s.rank%d' % (path, rank)


def _attach(logger, handler):
    handler.setLevel(logging.DEBUG)
    handler.setFormatter(_FORMAT)
    logger.addHandler(handler)


def setup_logger(output=None, name='passl'):
    logger = logging.getLogger(name)
    if name in logger_initialized:
        return logger
    rank = _rank()
    logger.propagate = False
    logger.setLevel(logging.INFO if rank == 0 else logging.WARNING)
    if rank == 0:
        _attach(logger, logging.StreamHandler(stream=sys.stdout))
    if output is not None:
        path = _log_path(output, rank)
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        _attach(logger, logging.FileHandler(path, mode='a'))
    logger_initialized.append(name)
    return logger


def get_logger(name='passl', output=None):
    if name in logger_initialized:
        return logging.getLogger(name)
    return setup_logger(output=output, name=name)
