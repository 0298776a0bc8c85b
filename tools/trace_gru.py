import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ['TG_GRU_DBG'] = str(16 | int(os.environ.get('DBG','0')))
import bench
from www2023tiger_amd import _lib
cfg = dict(bench.C2)
B,K,d = cfg['B'],cfg['K'],cfg['d']
stream = bench.make_stream(cfg['n_u'], cfg['n_i'], cfg['E'], cfg['T'], seed=0, d_e=d)
model,_ = bench.build_models(stream, d, K, 'left','left')
dev = torch.device('cuda:0')
res = tuple(torch.from_numpy(stream[k]).to(dev) for k in ('src','dst','neg','ts','eids'))
buf = model.StepBuffers(model, B, False, resident=res)
for _ in range(40): model.launch_step(buf)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
n = 240
out = np.zeros(n*4, dtype=np.uint64)
rc = raw.tg_debug_gru_trace(C.c_void_p(out.ctypes.data), n)
t = out.reshape(n,4).astype(np.int64)
t0 = t[:,0].min()
dur = t - t0
live = (t[:,2]-t[:,1]) > 1000
print('rc', rc, 'live blocks', live.sum())

for name, col in (('prologue', (t[:,1]-t[:,0])), ('loop', (t[:,2]-t[:,1])), ('epilogue', (t[:,3]-t[:,2])), ('total', t[:,3]-t[:,0])):
    v = col[live]/100.0
    print(name, 'us: mean %.2f min %.2f max %.2f' % (v.mean(), v.min(), v.max()))

