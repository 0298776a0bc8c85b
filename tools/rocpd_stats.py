#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ns, share) of a rocprofv3 rocpd database.
usage: tools/rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
scol = [r[1] for r in cur.execute(f'pragma table_info({sym})')]
name = 'display_name' if 'display_name' in scol else 'kernel_name'
rows = list(cur.execute(f'select s.{name}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) '
                        f'from {kd} d join {sym} s on d.kernel_id = s.id group by s.{name} order by 3 desc'))
tot = sum(r[2] for r in rows)
out = [('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs')]
for n, c, t, mn, mx in rows:
    out.append((n, c, t, round(t / c, 1), round(100.0 * t / tot, 3), mn, mx))
if len(sys.argv) > 2:
    csv.writer(open(sys.argv[2], 'w')).writerows(out)
for r in out[:45]:
    print('%-90s %6s %12s %10s %7s' % (str(r[0])[:90], r[1], r[2], r[3], r[4]))
