#!/usr/bin/env python3
"""Fixed-cost budget of the launches of the C2 streaming step (VERDICT r04 task 1, the fallback deliverable): in-kernel
s_memtime stamps per workgroup of ONE launch of the timed form - entry -> first operands requested -> k-loop done -> exit
- plus the launch's own span (first entry -> last exit) and the stagger of the workgroups' entries.
Needs a library with the stamps compiled in (they cost registers, so the production build leaves them out):
  make -C www2023tiger_amd/csrc clean && make -C www2023tiger_amd/csrc -j16 EXTRA='-DTG_PHASE_TRACE -DTG_CORE_TRACE'
usage: python tools/phase_budget.py fc1|fc2|qrows|updater|core|tile   (one launch kind per process: the library reads its
knobs once; tools/phase_budget.sh runs them all and prints the table)"""
import ctypes as C
import os
import sys

import numpy as np

which = sys.argv[1]
SEL = dict(fc1='172,1204', fc2='172,172', qrows='1032,172')
if which in SEL:
    os.environ['TG_GEMM_DBG'] = '16'
    os.environ['TG_PHASE_NK'] = SEL[which]
elif which == 'updater':
    os.environ['TG_GRU_DBG'] = '16'
elif which == 'core':
    os.environ['TG_CORE_DBG'] = '1'
elif which == 'tile':
    os.environ.update(TG_TILE_DBG='1', TG_GTAB='0', TG_ATTN_TILE='1')
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from www2023tiger_amd import _lib  # noqa: E402

c = bench.C2
B, nb = c['B'], 170
E = (nb + 2) * B
st = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=0, d_e=c['d'])
model, _ = bench.build_models(st, c['d'], c['K'], c['msg_src'], c['upd_src'])
model.fuse_attention()
model.eager_updates()
res = tuple(torch.from_numpy(st[k]).to(model.device) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
buf = model.StepBuffers(model, B, False, resident=res, prefetch=which != 'tile')
buf.io.lean = 1
for b in range(nb):
    model.launch_step(buf)
    if b == 100:
        cnt = buf.counts.tolist()
        model.note_rows(cnt[1], cnt[2])
torch.cuda.synchronize()
# ticks per microsecond: the same counter around a device-side wait of known length
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
raw = C.CDLL(_lib.LIB_PATH)


def stats(name, v):
    v = np.asarray(v, dtype=np.float64)
    return f'{name:34s} median {np.median(v):9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f}'


if which in ('fc1', 'fc2', 'qrows', 'updater'):
    n = 4096 if which != 'updater' else 2048
    t = np.zeros(n * 4, dtype=np.uint64)
    fn = raw.tg_debug_gemm_trace if which != 'updater' else raw.tg_debug_gru_trace
    assert fn(C.c_void_p(t.ctypes.data), n) == 0
    t = t.reshape(n, 4).astype(np.int64)
    t = t[:256]  # the step's launches are 256 persistent workgroups (higher slots hold stamps of one-off launches, e.g. table builds)
    t = t[t[:, 0] > 0]
    print(f'{which}: {len(t)} workgroups stamped (wavefront 0 of each, last launch of the run); s_memtime ticks (shader clock, ~2.1 GHz;'
          ' the counters of different XCDs are not aligned: only differences inside a workgroup are meaningful)')
    print(stats('entry -> first operands requested', t[:, 1] - t[:, 0]))
    print(stats('k-loop (waits for operands + MFMAs)', t[:, 2] - t[:, 1]))
    print(stats('fold / epilogue / stores', t[:, 3] - t[:, 2]))
    print(stats('workgroup lifetime', t[:, 3] - t[:, 0]))
elif which == 'core':
    nw = 3 * B
    t = np.zeros(nw * 5, dtype=np.uint64)
    assert raw.tg_debug_core_trace(C.c_void_p(t.ctypes.data), nw) == 0
    t = t.reshape(nw, 5).astype(np.int64)
    print(f'core: {nw} wavefronts (one centre each); s_memtime ticks')
    for k, nme in enumerate(['entry -> lists arrived', 'lists -> first key reduced', 'first key -> keys done', 'keys done -> exit']):
        print(stats(nme, t[:, k + 1] - t[:, k]))
    print(stats('wavefront lifetime', t[:, 4] - t[:, 0]))
elif which == 'tile':
    nbk = 512
    t = np.zeros(nbk * 16 * 8, dtype=np.uint64)
    assert raw.tg_debug_tile_trace(t.ctypes.data_as(C.c_void_p), nbk) == 0
    t = t.reshape(nbk, 16, 8).astype(np.int64)
    live = t[:, :, 0] > 0
    blocks = live.any(1)
    t, live = t[blocks], live[blocks]
    print(f'tile (one-launch attention k_attn_tile: G product + core + fc1 + fc2 per 16 centres): {len(t)} workgroups; s_memtime ticks')
    for i, nme in enumerate(['P0 centre rows', 'P1 G product', 'P2 core (own centres)', 'P2 wait at barrier', 'P3 fc1 product', 'P4 fc2 + store']):
        print(stats(nme, (t[:, :, i + 1] - t[:, :, i])[live]))
    print(stats('workgroup lifetime', (t[:, :, 6] - t[:, :, 0])[live]))
