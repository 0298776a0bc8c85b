#!/usr/bin/env python3
"""Where the time of the restart-mode evaluation pass goes (eval_edge_prediction(restart_mode=True), seq restarter, bs 200):
cProfile of the host side + wall time per batch."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from www2023tiger_amd.data.data_loader import BatchLoader, GraphCollator, InteractionData  # noqa: E402
from www2023tiger_amd.eval_utils import eval_edge_prediction  # noqa: E402

c = dict(bench.WORKLOADS['c2'])
bs, nb = 200, (int(sys.argv[1]) if len(sys.argv) > 1 else 100)
n = nb * bs
st = bench.make_stream(c['n_u'], c['n_i'], max(c['E'], n), c['T'], seed=0, d_e=c['d'])
model, _ = bench.build_models(st, c['d'], c['K'], c['msg_src'], c['upd_src'], restarter='seq', hist_len=40, dropout=0.1)
model.eval()
coll = GraphCollator(model.graph, c['K'], 1, restarter='seq', hist_len=40)
rs = np.random.RandomState(1)
lo = len(st['src']) - n  # the LAST n events: the restarted nodes have histories (from the first event on they would all be empty)
ev = InteractionData(st['src'][lo:], st['dst'][lo:], st['ts'][lo:], st['eids'][lo:], np.zeros(n, dtype=np.int64), seed=0, eval=True,
                     neg_dst=rs.randint(c['n_u'] + 1, c['n_u'] + c['n_i'] + 1, n))
dl = BatchLoader(ev, bs, coll)
for _ in range(2):
    model.reset()
    t0 = time.perf_counter()
    eval_edge_prediction(model, dl, model.device, restart_mode=True, uptodate_nodes=set())
    torch.cuda.synchronize()
    print(f'{(time.perf_counter() - t0) / nb * 1e3:.3f} ms per batch')
model.reset()
pr = cProfile.Profile()
pr.enable()
eval_edge_prediction(model, dl, model.device, restart_mode=True, uptodate_nodes=set())
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
pstats.Stats(pr).sort_stats('cumtime').print_stats(45)
