"""Same-box A/B of whole training steps: runs ``bench.py`` once per variant and repetition, interleaved, and prints the step
times side by side.  A variant is a label and a set of environment variables (and / or extra bench.py flags).

    python tools/step_ab.py [--workload moco] [--reps 3] [--steps 20] \
        base: \
        grid512: PASSL_WGRAD_TARGET_BLOCKS=512 PASSL_WGRAD_HALO_TARGET_BLOCKS=512 \
        dp: --dp-force

Why it exists: kernels that run on the side stream NEXT TO the main chain — weight gradients, the
collectives' issue stream — have a different optimum inside the step than in a stand-alone timing (tools/kbench), and the
boxes of a pool differ by more than most effects (3 % in round 6): only variants interleaved on ONE box compare.  One
gpurun call = one box: `gpurun -- 'python tools/step_ab.py ...'`."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_variants(words):
    out, cur = [], None
    for w in words:
        if w.endswith(':') and '=' not in w:
            cur = {'label': w[:-1], 'env': {}, 'flags': []}
            out.append(cur)
        elif cur is None:
            raise SystemExit('step_ab: a variant starts with "label:" (got %r)' % w)
        elif w.startswith('--'):
            cur['flags'].append(w)
        elif '=' in w:
            k, v = w.split('=', 1)
            cur['env'][k] = v
        else:
            cur['flags'].append(w)          # the value of the preceding flag
    return out


def run_once(v, args):
    env = dict(os.environ)
    env.update(v['env'])
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', args.workload, '--steps', str(args.steps),
           '--warmup', str(args.warmup), '--no-cpu-baseline', '--no-kernel-timing'] + v['flags']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.timeout)
    for line in r.stdout.split('\n'):
        if line.startswith('{'):
            return json.loads(line)['ms_per_step']
    sys.stderr.write(r.stderr[-2000:])
    return float('nan')


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--workload', default='moco')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--timeout', type=int, default=600)
    ap.add_argument('variants', nargs=argparse.REMAINDER)
    args = ap.parse_args()
    variants = parse_variants(args.variants)
    if len(variants) < 1:
        raise SystemExit('step_ab: no variants')
    times = {v['label']: [] for v in variants}
    for rep in range(args.reps):
        for v in variants:                       # interleaved: drift of the box hits every variant alike
            t = run_once(v, args)
            times[v['label']].append(t)
            print('%s %-24s rep %d: %.3f ms' % (args.workload, v['label'], rep + 1, t), flush=True)
    base = variants[0]['label']
    mean = {k: sum(t) / len(t) for k, t in times.items()}
    print('\n%-24s %10s %10s %10s   vs %s' % ('variant', 'mean ms', 'min ms', 'max ms', base))
    for v in variants:
        t = times[v['label']]
        print('%-24s %10.3f %10.3f %10.3f   %+.3f ms (%+.2f %%)' % (
            v['label'], mean[v['label']], min(t), max(t), mean[v['label']] - mean[base],
            100 * (mean[v['label']] / mean[base] - 1)))


if __name__ == '__main__':
    main()
