"""s_memtime stamps of the eager-updater launch (k_gru<2,4>, with / without the K-split) at C2: python tools/trace_gru_eager.py"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ['TG_GRU_DBG'] = str(16 | int(os.environ.get('DBG', '0')))
import bench
from www2023tiger_amd import _lib
cfg = dict(bench.C2)
B, K, d = cfg['B'], cfg['K'], cfg['d']
E = 200 * B
stream = bench.make_stream(cfg['n_u'], cfg['n_i'], E, cfg['T'] * E / cfg['E'], seed=0, d_e=d)
model, _ = bench.build_models(stream, d, K, 'left', 'left')
model.fuse_attention(); model.eager_updates()
dev = torch.device('cuda:0')
res = tuple(torch.from_numpy(stream[k]).to(dev) for k in ('src', 'dst', 'neg', 'ts', 'eids'))
buf = model.StepBuffers(model, B, False, resident=res)
for _ in range(150): model.launch_step(buf)
torch.cuda.synchronize()
cnt = buf.counts.tolist()
model.note_rows(cnt[1], cnt[2])  # as bench.py does: the bound on the unique positive nodes selects the 32-row blocks
for _ in range(10): model.launch_step(buf)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
n = 512
out = np.zeros(n * 4, dtype=np.uint64)
rc = raw.tg_debug_gru_trace(C.c_void_p(out.ctypes.data), n)
t = out.reshape(n, 4).astype(np.int64)
# s_memtime counters are per XCD (blockIdx % 8) and not aligned with each other: the last launch's blocks are picked per
# XCD, and only differences of stamps of one block are compared across XCDs
live = np.zeros(n, dtype=bool)
for x in range(8):
    ix = np.arange(x, n, 8)
    ok = (t[ix, 2] - t[ix, 1]) > 1000
    if not ok.any():
        continue
    last = t[ix[ok], 0].max()
    live[ix[ok & (t[ix, 0] > last - 150000)]] = True
print('rc', rc, 'live blocks', live.sum(), 'counts', buf.counts.cpu().numpy())
for name, col in (('prologue', (t[:, 1] - t[:, 0])), ('loop', (t[:, 2] - t[:, 1])), ('epilogue', (t[:, 3] - t[:, 2])),
                  ('total', t[:, 3] - t[:, 0])):
    v = col[live]
    print(name, 'ticks: mean %.0f min %.0f max %.0f' % (v.mean(), v.min(), v.max()))
for x in range(8):
    ix = np.arange(x, n, 8)
    ix = ix[live[ix]]
    if len(ix):
        t0 = t[ix, 0].min()
        print('xcd', x, 'blocks', len(ix), 'start spread', int(t[ix, 0].max() - t0), 'last end', int(t[ix, 3].max() - t0))
