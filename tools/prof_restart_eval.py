import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from www2023tiger_amd.data.data_loader import BatchLoader, GraphCollator, InteractionData
from www2023tiger_amd.eval_utils import eval_edge_prediction
rst=sys.argv[1] if len(sys.argv) > 1 else 'static'; bs=200; nb=100
c = bench.C2
n = nb * bs
st = bench.make_stream(c['n_u'], c['n_i'], max(c['E'], n), c['T'], seed=0, d_e=c['d'])
model, _ = bench.build_models(st, c['d'], c['K'], c['msg_src'], c['upd_src'], restarter=rst, hist_len=40, dropout=0.1)
model.eval()
coll = GraphCollator(model.graph, c['K'], 1, restarter=rst, hist_len=40)
rs = np.random.RandomState(1)
ev = InteractionData(st['src'][:n], st['dst'][:n], st['ts'][:n], st['eids'][:n], np.zeros(n, dtype=np.int64), seed=0, eval=True,
                     neg_dst=rs.randint(c['n_u'] + 1, c['n_u'] + c['n_i'] + 1, n))
dl = BatchLoader(ev, bs, coll)
model.reset(); eval_edge_prediction(model, dl, model.device, restart_mode=True, uptodate_nodes=set())
model.reset()
pr = cProfile.Profile(); pr.enable()
eval_edge_prediction(model, dl, model.device, restart_mode=True, uptodate_nodes=set())
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
