"""Per-kernel MFMA-busy share from one rocprofv3 PMC pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE).

    python tools/pmc_mfma_summary.py <results.db> <steps> [title...]

MfmaUtil is rocprofv3's own derived metric (counters list of ROCm 7.2):
    reduce(SQ_VALU_MFMA_BUSY_CYCLES, sum) / (reduce(GRBM_GUI_ACTIVE, max) * SIMD_NUM) * 100
with SIMD_NUM = 256 CUs x 4 SIMDs = 1024 on MI355X: the share of SIMD-cycles in which the matrix pipe
was busy while the kernel ran.  100 % = every SIMD issuing MFMAs back to back = the dense peak."""
import re
import sqlite3
import sys

SIMD_NUM = 256 * 4


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(?:<.*?>)?)\(', name)
    return (m.group(1) if m else name)[:72]


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    con = sqlite3.connect(db)
    per = {}        # dispatch -> {counter: [values]}
    meta = {}
    for did, name, dur, cn, cv in con.execute(
            'select dispatch_id, name, duration, counter_name, counter_value from pmc_events'):
        per.setdefault(did, {}).setdefault(cn, []).append(cv)
        meta[did] = (short(name), dur)
    agg = {}
    for did, c in per.items():
        k, dur = meta[did]
        e = agg.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0])
        e[0] += 1
        e[1] += dur
        e[2] += sum(c.get('SQ_VALU_MFMA_BUSY_CYCLES', [0]))
        e[3] += max(c.get('GRBM_GUI_ACTIVE', [0]))
        e[4] += sum(c.get('SQ_BUSY_CYCLES', [0]))
    print('# ' + ' '.join(sys.argv[3:]))
    print('%-72s %7s %10s %10s %9s' % ('kernel', 'calls', 'ms/step', 'MfmaUtil%', 'eff_GHz'))
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot_busy = tot_act = 0.0
    for k, (n, dur, busy, act, sqb) in rows:
        tot_busy += busy
        tot_act += act
        if dur / steps / 1e6 < 0.02:
            continue
        util = 100.0 * busy / max(act * SIMD_NUM, 1.0)
        print('%-72s %7d %10.3f %10.2f %9.2f' % (k, n, dur / steps / 1e6, util, act / max(dur, 1.0)))
    print('# whole step: MfmaUtil %.2f %% of SIMD-cycles (all kernels, GRBM_GUI_ACTIVE-weighted)'
          % (100.0 * tot_busy / max(tot_act * SIMD_NUM, 1.0)))


if __name__ == '__main__':
    main()
