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
for _ in range(160): model.launch_step(buf)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
n = 512
out = np.zeros(n * 4, dtype=np.uint64)
rc = raw.tg_debug_gru_trace(C.c_void_p(out.ctypes.data), n)
t = out.reshape(n, 4).astype(np.int64)
live = (t[:, 2] - t[:, 1]) > 1000
live &= t[:, 0] > t[live, 0].max() - 200000   # the last launch only (stamps of earlier launches linger in other slots)
t0 = t[live, 0].min()
print('rc', rc, 'live blocks', live.sum(), 'counts', buf.counts.cpu().numpy())
for name, col in (('start', t[:, 0] - t0), ('prologue', (t[:, 1] - t[:, 0])), ('loop', (t[:, 2] - t[:, 1])), ('epilogue', (t[:, 3] - t[:, 2])),
                  ('total', t[:, 3] - t[:, 0]), ('end', t[:, 3] - t0)):
    v = col[live] / 100.0
    print(name, 'us: mean %.2f min %.2f max %.2f' % (v.mean(), v.min(), v.max()))

idx = np.nonzero(live)[0]
order = np.argsort(t[live, 0])
print('blockIdx xcd start end (hundreds of ticks):')
for k in order:
    b = idx[k]
    print(b, b % 8, round((t[b, 0] - t0) / 100.0, 1), round((t[b, 3] - t0) / 100.0, 1))
