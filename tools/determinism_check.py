#!/usr/bin/env python3
"""Is the streaming step bit-reproducible?  Two models with equal weights run the same C2 batches as eager launches;
the first batch whose embeddings / state differ is reported, with the tensor that differs first."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

cfg = bench.C2
B, K, d = cfg['B'], cfg['K'], cfg['d']
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 60
E = (nb + 2) * B
stream = bench.make_stream(cfg['n_u'], cfg['n_i'], E, cfg['T'] * E / cfg['E'], seed=0, d_e=d)
dev = torch.device('cuda:0')
resident = tuple(torch.from_numpy(stream[k]).to(dev) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
ms, bufs = [], []
for _ in range(2):
    m, _ = bench.build_models(stream, d, K, cfg['msg_src'], cfg['upd_src'])
    m.fuse_attention()
    m.eager_updates()
    b = m.StepBuffers(m, B, False, resident=resident)
    b.io.lean = 1
    ms.append(m)
    bufs.append(b)
hint = int(os.environ.get('HINT_AFTER', '20'))
for step in range(nb):
    for m, b in zip(ms, bufs):
        m.launch_step(b)
    torch.cuda.synchronize()
    if step == hint:
        for m, b in zip(ms, bufs):
            c = b.counts.tolist()
            m.note_rows(c[1], c[2])
    pairs = [('h', bufs[0].h, bufs[1].h), ('left', ms[0].left_memory.vals, ms[1].left_memory.vals),
             ('right', ms[0].right_memory.vals, ms[1].right_memory.vals), ('pending', ms[0]._pending, ms[1]._pending),
             ('mailbox', ms[0].msg_store.node_msg_vals, ms[1].msg_store.node_msg_vals)]
    bad = [(n, float((x - y).abs().max())) for n, x, y in pairs if not torch.equal(x, y)]
    if bad:
        print(f'batch {step}: differs: {bad}')
        n, x, y = next(p for p in pairs if p[0] == bad[0][0])
        idx = torch.nonzero((x != y).any(1)).flatten()
        print(f'  {n}: {idx.numel()} rows differ, first rows {idx[:8].tolist()}')
        sys.exit(1)
print(f'{nb} batches: embeddings, memories, pending rows and mailbox bit-identical between two eager runs')
