"""Timeline view of a rocprofv3 kernel trace (csv): per-stream busy time, GPU idle time and the kernels that run
alone, over the steady-state steps.   python tools/trace_timeline.py <kernel_trace.csv[.gz]> <steps_in_window> [marker]

`marker` is a kernel name fragment that occurs once per step (default: sgd_kernel); the window is the last
<steps_in_window> steps, from the end of the marker launch before them to the end of the last one."""
import csv
import gzip
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(?:<.*?>)?)\(', name)
    return (m.group(1) if m else name)[:64]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    marker = sys.argv[3] if len(sys.argv) > 3 else 'sgd_kernel'
    op = gzip.open if path.endswith('.gz') else open
    with op(path, 'rt') as f:
        rows = list(csv.DictReader(f))
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Stream_Id'], short(r['Kernel_Name']),
           int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) //
           max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])))
          for r in rows]
    ev.sort()
    marks = [e[1] for e in ev if marker in e[3]]
    t0, t1 = marks[-steps - 1], marks[-1]
    win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    span = (t1 - t0) / 1e6
    print('window: %d steps, %.3f ms per step, %d kernels per step' % (steps, span / steps, len(win) // steps))
    busy = defaultdict(float)
    for s, e, st, _, _ in win:
        busy[st] += (e - s) / 1e6
    for st in sorted(busy, key=lambda k: -busy[k]):
        print('stream %-4s busy %.3f ms per step (%d kernels)' % (st, busy[st] / steps,
                                                                    sum(1 for w in win if w[2] == st) // steps))
    # sweep: time with k kernels in flight, and who runs alone
    pts = []
    for i, (s, e, _, _, _) in enumerate(win):
        pts.append((s, 1, i))
        pts.append((e, -1, i))
    pts.sort()
    depth_time = defaultdict(float)
    alone = defaultdict(float)
    alone_small = defaultdict(float)
    live = set()
    prev = t0
    for t, d, i in pts:
        dt = (t - prev) / 1e6
        depth_time[len(live)] += dt
        if len(live) == 1:
            j = next(iter(live))
            alone[win[j][3]] += dt
            if win[j][4] < 256:
                alone_small[win[j][3]] += dt
        prev = t
        if d > 0:
            live.add(i)
        else:
            live.discard(i)
    for k in sorted(depth_time):
        print('%d kernels in flight: %.3f ms per step' % (k, depth_time[k] / steps))
    print('kernels running alone (ms per step; [..] = of which with fewer than 256 workgroups):')
    for n in sorted(alone, key=lambda k: -alone[k])[:24]:
        print('  %-64s %.3f [%.3f]' % (n, alone[n] / steps, alone_small[n] / steps))
    # idle gaps by the kernel that follows
    gaps = defaultdict(float)
    end = t0
    for s, e, _, n, _ in win:
        if s > end:
            gaps[n] += (s - end) / 1e6
        end = max(end, e)
    print('idle before (ms per step):')
    for n in sorted(gaps, key=lambda k: -gaps[k])[:12]:
        print('  %-64s %.3f' % (n, gaps[n] / steps))
    # what surrounds the marker launch of the last step in the window (us relative to the marker's start)
    last = max(i for i, e in enumerate(win) if marker in e[3])
    m0 = win[last][0]
    print('around the last %s (start us rel. to it, duration us, stream, workgroups, kernel):' % marker)
    for s, e, st, n, wg in win[max(0, last - 16):last + 8]:
        print('  %10.1f %8.1f  s%-3s %6d  %s' % ((s - m0) / 1e3, (e - s) / 1e3, st, wg, n))


if __name__ == '__main__':
    main()
