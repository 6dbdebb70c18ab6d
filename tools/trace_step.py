#!/usr/bin/env python
"""Timeline of the LAST training step in a rocprofv3 kernel trace (steps are delimited by the k_adam launches):
    python tools/trace_step.py <..._kernel_trace.csv> [--all]
Prints start offset / duration / short kernel name / grid, the sum, and an aggregate per kernel template."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_adam' in r['Kernel_Name']]
a, b = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
t0 = int(rows[a]['Start_Timestamp'])
agg = collections.OrderedDict()
tot = 0
for r in rows[a:b]:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot += d
    n = r['Kernel_Name'].replace('void qk::(anonymous namespace)::', '').replace('qk::', '')
    n = re.sub(r'\(.*', '', n)[:64]
    if '--all' in sys.argv or d > 20000:
        print('%9.1f %8.1f  %-64s grid=%s' % ((int(r['Start_Timestamp']) - t0) / 1e3, d / 1e3, n, r.get('Grid_Size_X', '')))
    c = agg.setdefault(n, [0, 0])
    c[0] += 1; c[1] += d
print('kernels %d  sum %.1f us  span %.1f us' % (b - a, tot / 1e3, (int(rows[b - 1]['End_Timestamp']) - t0) / 1e3))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print('  %7.1f us  x%-3d %s' % (d / 1e3, c, n))
