#!/usr/bin/env python3
"""Where the host time of the reference-shaped evaluation loop goes (cProfile of eval_edge_prediction at bs 1024)."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from www2023tiger_amd.data.data_loader import BatchLoader, GraphCollator, InteractionData  # noqa: E402
from www2023tiger_amd.eval_utils import eval_edge_prediction  # noqa: E402

c = bench.C2
bs, nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 60
n = nb * bs
st = bench.make_stream(c['n_u'], c['n_i'], max(c['E'], n), c['T'], seed=0, d_e=c['d'])
model, _ = bench.build_models(st, c['d'], c['K'], c['msg_src'], c['upd_src'], restarter='static', dropout=0.1)
model.eval()
coll = GraphCollator(model.graph, c['K'], 1, restarter='static', hist_len=1)
rs = np.random.RandomState(1)
ev = InteractionData(st['src'][:n], st['dst'][:n], st['ts'][:n], st['eids'][:n], np.zeros(n, dtype=np.int64), seed=0, eval=True,
                     neg_dst=rs.randint(c['n_u'] + 1, c['n_u'] + c['n_i'] + 1, n))
dl = BatchLoader(ev, bs, coll)
for _ in range(2):
    model.reset()
    eval_edge_prediction(model, dl, model.device, restart_mode=False)
torch.cuda.synchronize()
model.reset()
import time
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
eval_edge_prediction(model, dl, model.device, restart_mode=False)
torch.cuda.synchronize()
pr.disable()
print(f'{(time.perf_counter() - t0) / nb * 1e3:.3f} ms per batch under the profiler')
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
