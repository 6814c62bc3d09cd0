"""Launcher — reference tools_v110/train.py:29-57.

    python tools/train.py -c configs/moco/moco_v2_r50.yaml -o dataloader.train.dataset.name=SyntheticTwoView
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py -c ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from passl_amd.engine.trainer import Trainer          # noqa: E402
from passl_amd.utils.config import get_config         # noqa: E402
from passl_amd.utils.options import parse_args        # noqa: E402
from passl_amd.utils.setup import setup               # noqa: E402


def main(args, cfg):
    setup(args, cfg)
    if args.dtype:
        cfg.compute_dtype = args.dtype
    trainer = Trainer(cfg)
    # continue training or evaluate: the checkpoint holds epoch and optimizer state
    if args.resume:
        trainer.resume(args.resume)
    # evaluate or fine-tune: only the model weights
    elif args.load:
        trainer.load(args.load)
    if args.evaluate_only:
        trainer.val()
        return
    if args.export:
        raise NotImplementedError('--export (paddle.jit inference model) is outside the hot path')
    trainer.train()


if __name__ == '__main__':
    args = parse_args()
    cfg = get_config(args.config_file, args.override)
    main(args, cfg)
