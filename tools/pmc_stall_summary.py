"""Per-dispatch stall breakdown from a rocprofv3 PMC pass with SQ counters, grouped by (kernel, grid).
   python tools/pmc_stall_summary.py <results.db>"""
import re, sqlite3, sys
def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name); name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(?:<.*?>)?)\(', name); return (m.group(1) if m else name)[:60]
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(pmc_events)")]
rows = {}
for did, name, dur, cn, cv in con.execute('select dispatch_id, name, duration, counter_name, counter_value from pmc_events'):
    e = rows.setdefault(did, {'name': short(name), 'dur': dur})
    e[cn] = e.get(cn, 0.0) + cv
agg = {}
order = []
for did in sorted(rows):
    e = rows[did]
    if not any(s in e['name'] for s in ('igemm', 'wgrad')):
        continue
    key = (e['name'], round(e.get('SQ_WAVES', 0)))
    if key not in agg:
        agg[key] = {'n': 0}; order.append(key)
    a = agg[key]; a['n'] += 1
    for k, v in e.items():
        if k != 'name':
            a[k] = a.get(k, 0.0) + v
ctrs = [c for c in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS',
                    'SQ_LDS_BANK_CONFLICT', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_LDS',
                    'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_VALU', 'SQ_LDS_IDX_ACTIVE', 'SQ_INST_CYCLES_VMEM_RD')
        if any(c in a for a in agg.values())]
print('%-62s %4s %8s | shares of SQ_WAVE_CYCLES: %s | mfma_busy/(4*wave_quadcyc)' % ('kernel', 'n', 'us', ' '.join(c.replace('SQ_', '')[:14] for c in ctrs[1:])))
for key in order:
    a = agg[key]; wc = max(a.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    print('%-62s %4d %8.1f | %s' % (key[0], a['n'], a['dur'] / a['n'] / 1e3,
          ' '.join('%14.3f' % (a.get(c, 0.0) / wc) for c in ctrs[1:])))
