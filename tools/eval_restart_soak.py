#!/usr/bin/env python3
"""Soak of the restart-mode evaluation forms on the C2-shaped stream (seq restarter, hist 40, bs 200): the last N batches
through tg_eval_restart_run at group 1 (the per-batch calls on two streams) against the host-sequenced one-stream pipeline -
scores, up-to-date set and final state must agree bit for bit - and at the default group (to rounding).
usage: tools/eval_restart_soak.py [n_batches=600]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from www2023tiger_amd.data.data_loader import BatchLoader, GraphCollator, InteractionData  # noqa: E402
from www2023tiger_amd.eval_utils import eval_edge_prediction  # noqa: E402

c = dict(bench.WORKLOADS['c2'])
bs, nb = 200, (int(sys.argv[1]) if len(sys.argv) > 1 else 600)
n = nb * bs + 77
st = bench.make_stream(c['n_u'], c['n_i'], max(c['E'], n), c['T'], seed=0, d_e=c['d'])
model, _ = bench.build_models(st, c['d'], c['K'], c['msg_src'], c['upd_src'], restarter='seq', hist_len=40, dropout=0.1)
model.eval()
coll = GraphCollator(model.graph, c['K'], 1, restarter='seq', hist_len=40)
lo = len(st['src']) - n
ev = InteractionData(st['src'][lo:], st['dst'][lo:], st['ts'][lo:], st['eids'][lo:], np.zeros(n, dtype=np.int64), seed=0, eval=True,
                     neg_dst=np.random.RandomState(1).randint(c['n_u'] + 1, c['n_u'] + c['n_i'] + 1, n))
out = {}
for form, env in (('group1', dict(TG_EVAL_RESTART_GROUP='1')), ('one_stream', dict(TG_EVAL_RESTART_RUN='0', TG_EVAL_RESTART_OVERLAP='0')),
                  ('default', {})):
    for k in ('TG_EVAL_RESTART_GROUP', 'TG_EVAL_RESTART_RUN', 'TG_EVAL_RESTART_OVERLAP'):
        os.environ.pop(k, None)
    os.environ.update(env)
    model.reset()
    up = set()
    t0 = time.perf_counter()
    res = eval_edge_prediction(model, BatchLoader(ev, bs, coll), model.device, restart_mode=True, uptodate_nodes=up)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[form] = (res, sorted(up), model.left_memory.vals.clone(), model.right_memory.vals.clone(), model.msg_store.node_msg_vals.clone(),
                 model.msg_store.has_msg_mask().clone())
    print(f'{form}: AP {res[0]:.6f} AUC {res[1]:.6f}, {len(up)} nodes restarted, {dt / nb * 1e3:.3f} ms per batch', flush=True)
a, b, d = out['group1'], out['one_stream'], out['default']
assert a[0] == b[0] and a[1] == b[1] and all(torch.equal(x, y) for x, y in zip(a[2:], b[2:])), 'group 1 differs from the one-stream pipeline'
assert a[1] == d[1] and torch.equal(a[-1], d[-1])
worst = max(float((x - y).abs().max() / x.abs().max().clamp_min(1e-30)) for x, y in zip(a[2:5], d[2:5]))
print(f'group 1 == one stream bit for bit; default group: same lists and bits, state within {worst:.2e} (relative to the largest entry), '
      f'AP / AUC differ by {abs(a[0][0] - d[0][0]):.1e} / {abs(a[0][1] - d[0][1]):.1e}')
