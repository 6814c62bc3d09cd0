"""Command line of tools/train.py (reference: passl_v110/utils/options.py:18-82)."""
import argparse


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='PASSL (MI355X-native engine)')
    parser.add_argument('-c', '--config-file', metavar='FILE', help='config file path')
    parser.add_argument('-o', '--override', action='append', default=[],
                        help='config options to be overridden')
    parser.add_argument('--resume', type=str, default=None, help='checkpoint path to resume')
    parser.add_argument('--load', type=str, default=None, help='checkpoint path to load')
    parser.add_argument('--pretrained', type=str, default=None, help='pretrained weights to load')
    parser.add_argument('--evaluate-only', action='store_true', help='skip training, evaluate only')
    parser.add_argument('--export', type=str, default=None, help='checkpoint to export')
    parser.add_argument('--num-gpus', type=int, default=1)
    parser.add_argument('--seed', type=int, default=None, help='fix random numbers by setting seed')
    parser.add_argument('--device', type=str, default=None, help='override cfg.device (gpu|cpu)')
    parser.add_argument('--dtype', type=str, default=None, choices=['bf16', 'fp32'],
                        help='compute dtype of the HIP path (default bf16)')
    return parser.parse_args(argv)
