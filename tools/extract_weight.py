"""Extract the weights under one prefix of a checkpoint — reference tools_v110/extract_weight.py:18-56
(same CLI: ``checkpoint --prefix backbone [--remove_prefix] --output out.pd|.pdparams``).

Checkpoints are the pickle-of-numpy files written by CheckpointHook (``{'epoch', 'state_dict',
'optimizer', 'lr_scheduler'}``) — the layout of the reference's own ``save`` helper
(passl_v110/hooks/checkpoint_hook.py:23-50), which is also what ``paddle.save`` produces for a
dygraph state_dict, so files move between the two code bases without conversion: state_dict keys
and logical shapes are the reference's (conv [Cout,Cin,kh,kw], Linear [in,out], BN
``_mean/_variance``)."""
import argparse
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='This script extracts weights from a checkpoint')
    parser.add_argument('checkpoint', help='checkpoint file')
    parser.add_argument('--prefix', type=str, default='backbone', help='destination file name')
    parser.add_argument('--remove_prefix', action='store_true',
                        help='remove prefix from keys of state dict')
    parser.add_argument('--output', type=str, help='destination file name')
    return parser.parse_args(argv)


def extract(ckpt, prefix='backbone', remove_prefix=False):
    output_dict = dict()
    has_prefix = False
    for key, value in ckpt['state_dict'].items():
        if key.startswith(prefix):
            if remove_prefix:
                key = key[len(prefix) + 1:]
            output_dict[key] = value
            has_prefix = True
    if not has_prefix:
        raise Exception('Cannot find a {} layer in the checkpoint.'.format(prefix))
    return output_dict


def main(argv=None):
    from passl_amd.utils.checkpoint import load_pickle
    args = parse_args(argv)
    assert (args.output.endswith('.pd') or args.output.endswith('.pdparams'))
    ckpt = load_pickle(args.checkpoint)
    out = extract(ckpt, args.prefix, args.remove_prefix)
    with open(args.output, 'wb') as f:
        pickle.dump(out, f, protocol=2)


if __name__ == '__main__':
    main()
