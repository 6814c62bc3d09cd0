"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch),
summed per training step.  FETCH_SIZE is doubled: on gfx950 rocprofv3 tallies the 128-byte
requests of wide coalesced reads at 64 B (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is
uncalibrated (reported as is).   python tools/pmc_summary.py <fetch.db> <write.db> <steps> [out.json]
With out.json: per-kernel bytes per launch (the figure bench.py reports as roofline.traffic)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(?:<.*?>)?)\(', name)
    return (m.group(1) if m else name)[:70]


def load(db):
    con = sqlite3.connect(db)
    out = {}
    for name, val, dur in con.execute('select name, counter_value, duration from pmc_events'):
        k = short(name)
        e = out.setdefault(k, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += val
        e[2] += dur
    return out


def main():
    f, w, steps = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
    print('%-70s %7s %12s %12s %10s %10s' % ('kernel', 'calls', 'fetch_GB/st', 'write_GB/st', 'ms/step', 'TB/s'))
    rows = []
    for k in f:
        calls, fkb, dur = f[k]
        wkb = w.get(k, [0, 0.0, 0.0])[1]
        fgb = 2.0 * fkb * 1024 / 1e9 / steps
        wgb = wkb * 1024 / 1e9 / steps
        ms = dur / 1e6 / steps
        rows.append((ms, k, calls, fgb, wgb))
    tf = tw = 0.0
    for ms, k, calls, fgb, wgb in sorted(rows, reverse=True)[:24]:
        print('%-70s %7d %12.3f %12.3f %10.3f %10.2f' % (k, calls, fgb, wgb, ms, (fgb + wgb) / max(ms, 1e-9)))
    for ms, k, calls, fgb, wgb in rows:
        tf += fgb
        tw += wgb
    print('# total per step: fetch %.2f GB, write %.2f GB' % (tf, tw))
    if len(sys.argv) > 4:
        import json
        per = {}
        for k in f:
            calls, fkb, dur = f[k]
            wkb = w.get(k, [0, 0.0, 0.0])[1]
            per[k] = {'launches': calls, 'fetch_bytes_per_launch': round(2.0 * fkb * 1024 / calls),
                      'write_bytes_per_launch': round(wkb * 1024 / calls),
                      'avg_launch_us': round(dur / 1e3 / calls, 2)}
        def klass(sel):
            ks = [k for k in per if sel(k)]
            n = sum(per[k]['launches'] for k in ks)
            return {'launches': n,
                    'hbm_bytes_per_launch': round(sum(per[k]['launches'] * (per[k]['fetch_bytes_per_launch'] +
                                                                              per[k]['write_bytes_per_launch'])
                                                      for k in ks) / max(n, 1)),
                    'kernels': sorted(ks)}
        classes = {'ring': klass(lambda k: 'igemm_ring_kernel' in k),
                   'g8p': klass(lambda k: 'igemm_8p_kernel' in k),
                   'igemm': klass(lambda k: k.startswith('igemm_kernel')),
                   'wgrad': klass(lambda k: 'wgrad' in k)}
        json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 '
                             '(gfx950 128-byte requests are tallied at 64 B), per launch',
                   'steps_profiled': steps, 'fetch_gb_per_step': round(tf, 2), 'write_gb_per_step': round(tw, 2),
                   'classes': classes, 'igemm_all_variants': klass(lambda k: 'igemm' in k), 'per_kernel': per},
                  open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()
