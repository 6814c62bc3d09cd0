"""Per-stream chain view of a rocprofv3 kernel trace (csv): for ONE steady-state step, what the kernels of each stream
cost in sequence — per kernel name: launches, time inside kernels, and the gaps BEFORE those launches (time during which
that stream runs nothing).  The stream whose busy + gap time fills the step is the critical chain.
    python tools/trace_chain.py <kernel_trace.csv[.gz]> [steps_in_window=4] [marker=sgd_kernel] [--list STREAM]"""
import csv
import gzip
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(?:<.*?>)?)\(', name)
    return (m.group(1) if m else name)[:60]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    path = args[0]
    steps = int(args[1]) if len(args) > 1 else 4
    marker = args[2] if len(args) > 2 else 'sgd_kernel'
    lst = sys.argv[sys.argv.index('--list') + 1] if '--list' in sys.argv else None
    op = gzip.open if path.endswith('.gz') else open
    with op(path, 'rt') as f:
        rows = list(csv.DictReader(f))
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Stream_Id'], short(r['Kernel_Name'])) for r in rows]
    ev.sort()
    marks = [e[1] for e in ev if marker in e[3]]
    t0, t1 = marks[-steps - 1], marks[-1]
    win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    print('window: %d steps, %.3f ms per step' % (steps, (t1 - t0) / 1e6 / steps))
    streams = defaultdict(list)
    for e in win:
        streams[e[2]].append(e)
    for st, evs in sorted(streams.items(), key=lambda kv: -sum(e[1] - e[0] for e in kv[1])):
        busy = sum(e[1] - e[0] for e in evs) / 1e3 / steps
        agg = defaultdict(lambda: [0, 0.0, 0.0])
        prev_end = None
        gaps_small = 0.0
        for s, e, _, name in evs:
            a = agg[name]
            a[0] += 1
            a[1] += (e - s) / 1e3
            if prev_end is not None:
                g = max(0, s - prev_end) / 1e3
                if g < 200.0:                 # a longer pause is the stream waiting for another one, not a launch gap
                    a[2] += g
                    gaps_small += g
            prev_end = max(prev_end or 0, e)
        print('\nstream %s: %d kernels per step, busy %.1f us, launch gaps (< 200 us each) %.1f us per step' % (
            st, len(evs) // steps, busy, gaps_small / steps))
        print('  %-60s %6s %10s %10s %10s' % ('kernel', 'calls', 'us/step', 'avg us', 'gap before'))
        for name, (n, t, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:28]:
            print('  %-60s %6.1f %10.1f %10.1f %10.1f' % (name, n / steps, t / steps, t / n, g / steps))
        if lst == st:
            last = [e for e in evs if e[0] >= marks[-2]]
            prev_end = None
            for s, e, _, name in last:
                print('    %10.1f %8.1f %8.1f  %s' % ((s - marks[-2]) / 1e3, (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3, name))
                prev_end = e


if __name__ == '__main__':
    main()
