#!/usr/bin/env python3
"""Does a memory-bound kernel overlap with an MFMA-bound product when the two are launched on two streams?  (The premise
of running the attention core beside fc1 at C5 shape.)  GEMM: 196 608 x 1 280 -> 256 (k_gemm_rb<2, 2>, two 228-register
blocks per CU); memory side: a row gather of N x 1 KB rows from a 10 M-row table (few registers).  Prints each alone, both in
series and both on two streams."""
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from www2023tiger_amd import hip_ops  # noqa: E402
from www2023tiger_amd.model.dense import linear_forward  # noqa: E402

dev = torch.device('cuda:0')
M, K, N = 196608, 1280, 256
x = torch.randn(M, K, device=dev)
lin = nn.Linear(K, N, device=dev)
table = torch.randn(10_000_001, 256, device=dev)
ids = torch.randint(0, 10_000_001, (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000,), device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def gemm():
    with torch.no_grad():
        linear_forward(lin, x)


def gather():
    hip_ops.gather_rows(table, ids)


def both():
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(s1):
        s1.wait_event(ev)
        gemm()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        gather()
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.current_stream().wait_stream(s2)


tg, tm = timed(gemm), timed(gather)
ts = timed(lambda: (gemm(), gather()))
tb = timed(both)
print(f'gemm {tg:.3f} ms, gather {tm:.3f} ms, in series {ts:.3f} ms, on two streams {tb:.3f} ms (ideal max {max(tg, tm):.3f})')
