import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from www2023tiger_amd.data.data_loader import BatchLoader, GraphCollator, InteractionData
from www2023tiger_amd import eval_utils
from www2023tiger_amd.model import training
c = bench.C2
bs = int(sys.argv[1]); nb = 300
n = nb * bs
st = bench.make_stream(c['n_u'], c['n_i'], max(c['E'], n), c['T'], seed=0, d_e=c['d'])
model, _ = bench.build_models(st, c['d'], c['K'], c['msg_src'], c['upd_src'], restarter='static', dropout=0.1)
model.eval()
coll = GraphCollator(model.graph, c['K'], 1, restarter='static', hist_len=1)
rs = np.random.RandomState(1)
ev = InteractionData(st['src'][:n], st['dst'][:n], st['ts'][:n], st['eids'][:n], np.zeros(n, dtype=np.int64), seed=0, eval=True,
                     neg_dst=rs.randint(c['n_u'] + 1, c['n_u'] + c['n_i'] + 1, n))
dl = BatchLoader(ev, bs, coll)
acc = {'t': 0.0, 'n': 0}
orig = training.TrainBuffers.launch
def timed(self, *a, **k):
    t0 = time.perf_counter(); r = orig(self, *a, **k); acc['t'] += time.perf_counter() - t0; acc['n'] += 1; return r
training.TrainBuffers.launch = timed
for rep in range(2):
    model.reset(); acc['t'] = 0.0; acc['n'] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eval_utils.eval_edge_prediction(model, dl, model.device, restart_mode=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'bs={bs}: wall {dt / nb * 1e6:.1f} us/batch, host inside launch() {acc["t"] / acc["n"] * 1e6:.1f} us/batch over {acc["n"]} launches')
