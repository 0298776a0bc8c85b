#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of k_gemm blocks for one torch-Linear-shaped product via tg_linear_fwd.
usage: TG_GEMM_DBG=16 python tools/trace_gemm.py M K N"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('TG_GEMM_DBG', '16')
from www2023tiger_amd._lib import TgLinear, check, lib, ptr  # noqa: E402

M, K, N = (int(x) for x in sys.argv[1:4])
dev = torch.device('cuda:0')
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev)
b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
lin = TgLinear(ptr(w), ptr(b))
raw = C.CDLL(lib._name)
for _ in range(5):
    check(lib.tg_linear_fwd(M, ptr(x), K, C.byref(lin), N, 0, ptr(out), None), 'lin')
torch.cuda.synchronize()
nb = min(4096, 8 * ((M + 63) // 64 + 7) // 8 * ((N + 63) // 64))
buf = np.zeros(4 * nb, dtype=np.uint64)
raw.tg_debug_gemm_trace(buf.ctypes.data_as(C.c_void_p), nb)
t = buf.reshape(nb, 4).astype(np.int64)
t = t[(t[:, 3] > 0) & (t[:, 0] > 0)]
t0 = t[:, 0].min()
pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
print(f'blocks {len(t)}  (s_memtime ticks; compare with the kernel duration to get the tick rate)')
print('prologue  mean %.0f  loop mean %.0f (%.0f per k-tile)  epilogue mean %.0f' % (pro.mean(), loop.mean(), loop.mean() / ((K + 31) // 32), epi.mean()))
print('first entry -> last exit: %d ticks; entry spread %d ticks; block lifetime mean %d' % (t[:, 3].max() - t0, t[:, 0].max() - t0, (t[:, 3] - t[:, 0]).mean()))
ref = x @ w.t() + b
print('max err', float((out - ref).abs().max()))
