#!/usr/bin/env python3
"""Throughput of the reference-style Python training loop on this package's drop-in API (BatchLoader ->
contrast_and_mutual_learning -> loss.backward() -> torch.optim.Adam.step()), next to FusedTrainer.
usage: python tools/loop_bench.py [--restarter seq|static] [--contrast-only]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from www2023tiger_amd.data.data_loader import BatchLoader, GraphCollator, InteractionData  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--restarter', default='static')
ap.add_argument('--contrast-only', action='store_true')
ap.add_argument('--steps', type=int, default=60)
ap.add_argument('--eval', type=int, default=0, help='time eval_edge_prediction with this batch size instead')
ap.add_argument('--adam', default='torch', choices=['torch', 'device-flags'])
ap.add_argument('--item', action='store_true', help='read the loss back every iteration, as the reference loop does')
args = ap.parse_args()
c = bench.C2
B = c['B']
E = (args.steps + 12) * B
st = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=0, d_e=c['d'])
model, _ = bench.build_models(st, c['d'], c['K'], c['msg_src'], c['upd_src'], restarter=args.restarter, hist_len=20,
                              dropout=0.1)
dev = model.device
data = InteractionData(st['src'], st['dst'], st['ts'], st['eids'], np.zeros(E, dtype=np.int64), seed=0, eval=False)
coll = GraphCollator(model.graph, c['K'], 1, restarter=args.restarter, hist_len=20)
dl = BatchLoader(data, B, coll)
from www2023tiger_amd import optim as tg_optim  # noqa: E402
opt = (torch.optim.Adam if args.adam == 'torch' else tg_optim.Adam)(model.parameters(), lr=1e-4)
if args.eval:
    from www2023tiger_amd.eval_utils import eval_edge_prediction
    n = args.steps * args.eval
    rs = np.random.RandomState(1)
    ev = InteractionData(st['src'][:n], st['dst'][:n], st['ts'][:n], st['eids'][:n], np.zeros(n, dtype=np.int64), seed=0,
                         eval=True, neg_dst=rs.randint(c['n_u'] + 1, c['n_u'] + c['n_i'] + 1, n))
    edl = BatchLoader(ev, args.eval, coll)
    model.eval()
    for rep in range(2):
        model.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ap_, auc_ = eval_edge_prediction(model, edl, dev, restart_mode=False)
        dt = time.perf_counter() - t0
    print(f'eval_edge_prediction bs={args.eval}: {dt / args.steps * 1e3:.3f} ms/batch, {n / dt / 1e6:.3f} M events/s, AP {ap_:.4f}')
    sys.exit(0)
model.train()
t0 = None
for i, (src, dst, neg, ts, eids, _, cg) in enumerate(dl):
    if i == 10:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    if i == 10 + args.steps:
        break
    src, dst, neg, eids = (x.long().to(dev) for x in (src, dst, neg, eids))
    ts = ts.float().to(dev)
    opt.zero_grad()
    c_loss, m_loss = model.contrast_and_mutual_learning(src, dst, neg, ts, eids, cg, contrast_only=args.contrast_only)
    loss = c_loss + m_loss
    loss.backward()
    opt.step()
    if args.item:
        last = loss.item()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'python loop ({args.adam} Adam, item={args.item}), {args.restarter} restarter, contrast_only={args.contrast_only}: {dt / args.steps * 1e3:.3f} ms/iteration, '
      f'{args.steps * B / dt / 1e6:.3f} M events/s, last loss {float(loss):.4f}')
