#!/usr/bin/env python3
"""Times tg_linear_fwd (k_gemm) on a list of shapes: python tools/gemm_shapes.py M,K,N [M,K,N ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from www2023tiger_amd import hip_ops  # noqa: E402
from www2023tiger_amd._lib import TgLinear, check, lib, ptr  # noqa: E402

dev = torch.device('cuda')
for spec in sys.argv[1:]:
    M, K, N = (int(x) for x in spec.split(','))
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    lin = TgLinear(ptr(w), ptr(b))
    run = lambda: check(lib.tg_linear_fwd(M, ptr(x), K, C.byref(lin), N, 1, ptr(out), hip_ops.stream_ptr(dev)), 'linear')
    for _ in range(5):
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
    ref = torch.relu(x @ w.T + b)
    err = float((out - ref).abs().max())
    print(f'M={M} K={K} N={N}: {us:.2f} us  {2 * M * K * N / us / 1e6:.1f} TF/s  max|err|={err:.2e}', flush=True)
