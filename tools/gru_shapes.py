#!/usr/bin/env python3
"""Times tg_gru_fwd (fused GRU cell) on a list of shapes: python tools/gru_shapes.py ROWS,XW,D [...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from www2023tiger_amd import hip_ops  # noqa: E402
from www2023tiger_amd._lib import check, lib, ptr  # noqa: E402

dev = torch.device('cuda')
for spec in sys.argv[1:]:
    n, xw, d = (int(v) for v in spec.split(','))
    x = torch.randn(n, xw, device=dev)
    h = torch.randn(n, d, device=dev)
    cell = torch.nn.GRUCell(xw, d).to(dev)
    out = torch.empty(n, d, device=dev)
    run = lambda: check(lib.tg_gru_fwd(n, ptr(x), xw, ptr(h), d, ptr(cell.weight_ih), ptr(cell.weight_hh), ptr(cell.bias_ih),
                                       ptr(cell.bias_hh), ptr(out), hip_ops.stream_ptr(dev)), 'gru')
    for _ in range(3):
        run()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 100
    with torch.no_grad():
        ref = cell(x, h)
    err = float((out - ref).abs().max())
    print(f'rows={n} xw={xw} d={d}: {us:.1f} us  {2.0 * n * 3 * d * (xw + d) / us / 1e6:.1f} TF/s  max|err|={err:.1e}', flush=True)
