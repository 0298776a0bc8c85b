#!/usr/bin/env python3
"""Timeline of ONE replayed step from a rocprofv3 rocpd database: every kernel between the last two launches of an anchor
kernel (default k_adam), with its start offset, duration and queue - shows what runs beside what on the side lane.
usage: tools/rocpd_timeline.py results.db [anchor-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_adam('
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
scol = [r[1] for r in cur.execute(f'pragma table_info({sym})')]
dcol = [r[1] for r in cur.execute(f'pragma table_info({kd})')]
name = 'display_name' if 'display_name' in scol else 'kernel_name'
q = 'queue_id' if 'queue_id' in dcol else ('stream_id' if 'stream_id' in dcol else 'tid')
rows = list(cur.execute(f'select s.{name}, d.start, d.end, d.{q} from {kd} d join {sym} s on d.kernel_id = s.id order by d.start'))
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
if len(idx) < 2:
    sys.exit('anchor kernel not found twice')
a, b = idx[-2], idx[-1]
t0 = rows[a][2]
short = lambda n: n.split('(')[0].replace('void ', '').replace('tg::', '')[:60]
queues = {}
print(f'step of {(rows[b][2] - rows[a][2]) / 1e3:.1f} us, {b - a} kernels; columns: start us, duration us, queue, kernel')
busy_end = {}
for n, s, e, qq in rows[a + 1:b + 1]:
    k = queues.setdefault(qq, len(queues))
    print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{k}  {"    " * k}{short(n)}')
