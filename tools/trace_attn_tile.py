#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of k_attn_tile's phases (per wavefront) on a C2-shaped (or --workload) stream.
usage: TG_TILE_DBG=1 python tools/trace_attn_tile.py [--workload c2] [--batches 30]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('TG_TILE_DBG', '1')
import bench  # noqa: E402
from www2023tiger_amd._lib import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='c2')
ap.add_argument('--batches', type=int, default=40)
args = ap.parse_args()
cfg = bench.WORKLOADS[args.workload]
B, K, d = cfg['B'], cfg['K'], cfg['d']
E = (args.batches + 2) * B
no_feats = bool(cfg.get('no_feats'))
stream = bench.make_stream(cfg['n_u'], cfg['n_i'], E, cfg['T'] * E / cfg['E'], seed=0, d_e=d,
                           integer_ts=cfg.get('integer_ts', True), with_efeats=not no_feats)
model, _ = bench.build_models(stream, d, K, cfg['msg_src'], cfg['upd_src'], zero_nfeats=not no_feats)
model.fuse_attention()
model.eager_updates()
dev = torch.device('cuda:0')
resident = tuple(torch.from_numpy(stream[k]).to(dev) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
buf = model.StepBuffers(model, B, False, resident=resident)
buf.io.lean = 1
for _ in range(args.batches):
    model.launch_step(buf)
torch.cuda.synchronize()
raw = C.CDLL(lib._name)
nb = 512
t = np.zeros(nb * 16 * 8, dtype=np.uint64)
assert raw.tg_debug_tile_trace(t.ctypes.data_as(C.c_void_p), nb) == 0
t = t.reshape(nb, 16, 8).astype(np.int64)
live = t[:, :, 0] > 0
blocks = live.any(1)
t = t[blocks]
live = live[blocks]
print(f'workgroups traced {len(t)}, wavefronts per workgroup {int(live[0].sum())}  (s_memtime ticks, 100 MHz => x10 ns)')
names = ['P0 centre rows', 'P1 G product', 'P2 core (own centres)', 'P2 wait at barrier', 'P3 fc1 product', 'P4 fc2 + store']
for i, n in enumerate(names):
    dts = (t[:, :, i + 1] - t[:, :, i])[live]
    print(f'  {n:26s} mean {dts.mean():8.1f}  min {dts.min():6d}  max {dts.max():6d}')
life = (t[:, :, 6] - t[:, :, 0])[live]
print(f'  workgroup lifetime mean {life.mean():.1f} max {life.max()}; first entry -> last exit {t[:, :, 6][live].max() - t[:, :, 0][live].min()}; '
      f'entry spread {t[:, :, 0][live].max() - t[:, :, 0][live].min()}')
