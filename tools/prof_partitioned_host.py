"""Host-side profile of the partitioned multi-GPU step with one rank (cProfile over 300 steps): python tools/prof_partitioned_host.py"""
import cProfile, os, pstats, socket, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as tdist
import bench
from www2023tiger_amd import dist as D
with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
tdist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
c = bench.C2; B, n_steps = c['B'], 500
E = (n_steps + 2) * B
stream = bench.make_stream(c['n_u'], c['n_i'], E, c['T'] * E / c['E'], seed=0, d_e=c['d'])
model, _ = bench.build_models(stream, c['d'], c['K'], c['msg_src'], c['upd_src'], restarter='static', device='cuda:0')
model.fuse_attention()
owner = D.balanced_owner_table(stream['n_nodes'], stream['dst'], 1)
rs = D.ResidentPartitionedStream(model, stream, owner, 0, 1, B, n_steps, physical=True)
for _ in range(150): rs.step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): rs.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
tdist.destroy_process_group()
