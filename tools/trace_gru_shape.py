#!/usr/bin/env python3
"""Per-block s_memtime stamps (prologue / loop / epilogue) of the GRU kernel tg_gru_fwd picks for a shape:
python tools/trace_gru_shape.py ROWS XW D   (sets TG_GRU_DBG=16; diagnostic only)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ['TG_GRU_DBG'] = '16'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from www2023tiger_amd import _lib, hip_ops  # noqa: E402
from www2023tiger_amd._lib import check, lib, ptr  # noqa: E402

n, xw, d = (int(v) for v in sys.argv[1:4])
dev = torch.device('cuda')
x = torch.randn(n, xw, device=dev)
h = torch.randn(n, d, device=dev)
cell = torch.nn.GRUCell(xw, d).to(dev)
out = torch.empty(n, d, device=dev)
for _ in range(3):
    check(lib.tg_gru_fwd(n, ptr(x), xw, ptr(h), d, ptr(cell.weight_ih), ptr(cell.weight_hh), ptr(cell.bias_ih),
                         ptr(cell.bias_hh), ptr(out), hip_ops.stream_ptr(dev)), 'gru')
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
nb = 2048
buf = np.zeros(nb * 4, dtype=np.uint64)
raw.tg_debug_gru_trace(C.c_void_p(buf.ctypes.data), nb)
t = buf.reshape(nb, 4).astype(np.int64)
live = (t[:, 3] > t[:, 0]) & (t[:, 2] - t[:, 1] > 100)
t = t[live]
print('traced blocks', len(t), '(counter ticks; 100 MHz constant clock => x10 ns)')
for name, col in (('prologue', t[:, 1] - t[:, 0]), ('loop', t[:, 2] - t[:, 1]), ('epilogue', t[:, 3] - t[:, 2]),
                  ('total', t[:, 3] - t[:, 0])):
    print(f'{name:9s} mean {col.mean():9.1f}  min {col.min():7d}  max {col.max():7d}')
span = t[:, 3].max() - t[:, 0].min()
print('first entry -> last exit of the traced blocks:', span)
