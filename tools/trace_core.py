#!/usr/bin/env python3
"""Per-wavefront s_memtime stamps of the attention core (k_attn_core) in the C2 streaming step:
entry -> lists arrived -> first key reduced -> keys done -> exit.  Needs a library built with the stamps compiled in:
  make -C www2023tiger_amd/csrc clean && make -C www2023tiger_amd/csrc CXXFLAGS+=-DTG_CORE_TRACE   (rebuild without it afterwards:
  the stamps cost the production kernel its third wavefront per SIMD);  python tools/trace_core.py  (sets TG_CORE_DBG=1)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ['TG_CORE_DBG'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from www2023tiger_amd import _lib  # noqa: E402

c = bench.C2
B = c['B']
nb = 160
E = (nb + 2) * B
st = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=0, d_e=c['d'])
model, _ = bench.build_models(st, c['d'], c['K'], c['msg_src'], c['upd_src'])
model.fuse_attention()
model.eager_updates()
res = tuple(torch.from_numpy(st[k]).to(model.device) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
buf = model.StepBuffers(model, B, False, resident=res, prefetch=True)
buf.io.lean = 1
for b in range(nb):
    model.launch_step(buf)
    if b == 100:
        cnt = buf.counts.tolist()
        model.note_rows(cnt[1], cnt[2])
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
nw = 3 * B
t = np.zeros(nw * 5, dtype=np.uint64)
raw.tg_debug_core_trace(C.c_void_p(t.ctypes.data), nw)
t = t.reshape(nw, 5).astype(np.int64)
t0 = t[:, 0].min()
print('waves', nw, ' (s_memtime ticks since the first wavefront entered; the counter runs at ~100 MHz or at the shader clock, see span)')
names = ['entry', 'lists arrived', 'first key reduced', 'keys done', 'exit']
for k, n in enumerate(names):
    col = t[:, k] - t0
    print(f'{n:18s} min {col.min():8d}  median {int(np.median(col)):8d}  p90 {int(np.percentile(col, 90)):8d}  max {col.max():8d}')
d = np.diff(t, axis=1)
for k, n in enumerate(['entry->lists', 'lists->first key', 'first key->keys done', 'keys done->exit']):
    print(f'{n:22s} median {int(np.median(d[:, k])):8d}  p90 {int(np.percentile(d[:, k], 90)):8d}')
