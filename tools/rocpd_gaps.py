#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 rocpd database, by (previous kernel -> next kernel) pair, over the
region where one pair repeats most (the replayed step).  usage: tools/rocpd_gaps.py results.db"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
scol = [r[1] for r in cur.execute(f'pragma table_info({sym})')]
name = 'display_name' if 'display_name' in scol else 'kernel_name'
rows = list(cur.execute(f'select s.{name}, d.start, d.end from {kd} d join {sym} s on d.kernel_id = s.id order by d.start'))
short = lambda n: n.split('(')[0].replace('void ', '').replace('tg::', '')[:48]
gaps = collections.defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    if s1 - e0 < 50000:  # inside a chain (not across host pauses)
        gaps[(short(n0), short(n1))].append(s1 - e0)
print('%-50s -> %-50s %6s %9s %9s' % ('previous kernel', 'next kernel', 'n', 'mean ns', 'median'))
for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:14]:
    v.sort()
    print('%-50s -> %-50s %6d %9.0f %9d' % (a, b, len(v), sum(v) / len(v), v[len(v) // 2]))
