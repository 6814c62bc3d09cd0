"""Per-kernel summary of a rocprofv3 rocpd database (the `top_kernels` view), one line per kernel,
with ms per training step.   python tools/rocpd_summary.py <results.db> <steps> [title...]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(?:<.*?>)?)\(', name)
    n = m.group(1) if m else name
    return n[:78]


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    con = sqlite3.connect(db)
    rows = con.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    print('# ' + ' '.join(sys.argv[3:]))
    print('%-78s %8s %12s %9s %6s %11s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct', 'ms_per_step'))
    tot = 0.0
    for name, calls, total, avg, pct in rows:
        tot += total
        print('%-78s %8d %12.1f %9.2f %6.2f %11.3f' % (short(name), calls, total, avg, pct, total / steps / 1e3))
    print('# sum of kernel time per step: %.2f ms' % (tot / steps / 1e3))


if __name__ == '__main__':
    main()
