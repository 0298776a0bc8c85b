#!/usr/bin/env python3
"""What the vendor library (rocBLAS / hipBLASLt through torch.matmul) reaches on the C5-shaped products, as a ceiling for the
hand-written f32 MFMA kernels: fc1 196 608 x 1 280 -> 256, updater-sized 81 920 x 1 024 -> 768, query rows 81 920 x 256 -> 1 024."""
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from www2023tiger_amd.model.dense import linear_forward  # noqa: E402

dev = torch.device('cuda:0')
torch.backends.cuda.matmul.allow_tf32 = False


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for M, K, N in ((196608, 1280, 256), (81920, 1024, 768), (81920, 256, 1024), (196608, 256, 256)):
    x = torch.randn(M, K, device=dev)
    lin = nn.Linear(K, N, device=dev)
    out = torch.empty(M, N, device=dev)
    with torch.no_grad():
        t_lib = timed(lambda: torch.addmm(lin.bias, x, lin.weight.t(), out=out))
        t_own = timed(lambda: linear_forward(lin, x))
    fl = 2.0 * M * K * N
    print(f'{M} x {K} -> {N}: vendor {t_lib:.3f} ms = {fl / t_lib / 1e9:.1f} TF/s; own {t_own:.3f} ms = {fl / t_own / 1e9:.1f} TF/s')
