#!/usr/bin/env python3
"""Exchange period k of the replicated multi-GPU layout (www2023tiger_amd.dist.PeriodicShardedRunner; SURVEY.md s8 e item 4):
what staleness costs.  R ranks (processes sharing this box's one GPU, gloo) stream the C2-shaped synthetic stream with
period k in {1, 2, 4, 8}; k = 1 is the exact engine.  Reported per k: the largest and the mean row-relative difference of
the embeddings h(t-) of all events against k = 1, AP / AUC of the link scores (score head on (h_src, h_dst) against
(h_src, h_neg), hit term left out; random-initialised weights, so AP / AUC sit near 0.5 and only their DIFFERENCE between
periods means anything), rows sent per rank and batch.  One JSON line on stdout.
usage: python tools/period_drift.py [ranks=2] [batches=48] [events per global batch=2048]"""
import json
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PERIODS = (1, 2, 4, 8)


def worker(rank, world, port, n_batches, Bg, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    from www2023tiger_amd.dist import HipBackend, PeriodicShardedRunner, balanced_owner_table
    from www2023tiger_amd.eval_utils import ap_auc_windows
    c = dict(bench.WORKLOADS['c2'])
    E = n_batches * Bg
    st = bench.make_stream(c['n_u'], c['n_i'], max(E, 4096), c['T'] * max(E, 4096) / c['E'], seed=0, d_e=c['d'])
    owner = balanced_owner_table(st['n_nodes'], st['dst'][:E], world)
    res = {}
    for k in PERIODS:
        model, _ = bench.build_models(st, c['d'], c['K'], c['msg_src'], c['upd_src'], restarter='static', device='cuda:0')
        backend = HipBackend(model, cap=Bg)
        runner = PeriodicShardedRunner(backend, owner, rank, world, cap=Bg, period=k)
        emb = np.zeros((E, 3, c['d']), dtype=np.float32)  # src, dst, neg embeddings of the events this rank embedded
        mine = np.zeros(E, dtype=bool)
        for b in range(n_batches):
            sl = slice(b * Bg, (b + 1) * Bg)
            a = [st[x][sl] for x in ('src', 'dst', 'neg', 'ts', 'eids')]
            li = np.nonzero(owner[a[1]] == rank)[0]  # (unbalanced plan: an event runs on the owner of its destination)
            runner.step(*a)
            n = len(li)
            h = backend.buf.h[:3 * n].cpu().numpy().reshape(3, n, c['d'])
            emb[b * Bg + li] = h.transpose(1, 0, 2)
            mine[b * Bg + li] = True
        runner.flush()
        backend.check_invariants()
        with torch.no_grad():
            x = torch.from_numpy(emb[mine]).to('cuda:0')
            pos = model.score_fn(x[:, 0].contiguous(), x[:, 1].contiguous()).reshape(-1)
            neg = model.score_fn(x[:, 0].contiguous(), x[:, 2].contiguous()).reshape(-1)
        res[k] = dict(emb=emb, mine=mine, pos=pos.cpu().numpy(), neg=neg.cpu().numpy(), sent=runner.exchanged_rows)
        del model, backend, runner
        torch.cuda.empty_cache()
    out = {}
    ref = res[1]
    for k in PERIODS:
        r = res[k]
        e, e1 = r['emb'][r['mine']].reshape(-1, c['d']), ref['emb'][ref['mine']].reshape(-1, c['d'])
        nrm = np.linalg.norm(e1, axis=1)
        rel = np.linalg.norm(e - e1, axis=1) / np.maximum(nrm, 1e-3)
        out[k] = dict(max_abs=float(np.abs(e - e1).max()), max_abs_over_max_ref=float(np.abs(e - e1).max() / np.abs(e1).max()),
                      row_rel_max=float(rel.max()), row_rel_mean=float(rel.mean()), row_rel_p99=float(np.quantile(rel, 0.99)),
                      rows_sent_per_batch=r['sent'] / n_batches, events=int(r['mine'].sum()),
                      pos=r['pos'], neg=r['neg'])
    gathered = [None] * world
    tdist.all_gather_object(gathered, {k: {kk: vv for kk, vv in v.items()} for k, v in out.items()})
    if rank == 0:
        line = dict(what='exchange period k of the replicated layout against k = 1 (exact): embedding drift and link-score AP / AUC',
                    ranks=world, batches=n_batches, events_per_global_batch=Bg, workload=c['name'],
                    note='processes share ONE GPU over gloo: a rehearsal of the algorithm, no timing; untrained weights', periods={})
        for k in PERIODS:
            pos = torch.from_numpy(np.concatenate([g[k]['pos'] for g in gathered])).to('cuda:0').sigmoid()
            neg = torch.from_numpy(np.concatenate([g[k]['neg'] for g in gathered])).to('cuda:0').sigmoid()
            ap, auc, _ = ap_auc_windows(pos, neg, 200)
            line['periods'][str(k)] = dict(
                ap=float(ap.mean().item()), auc=float(auc.mean().item()),
                row_rel_max=max(g[k]['row_rel_max'] for g in gathered), row_rel_p99=max(g[k]['row_rel_p99'] for g in gathered),
                row_rel_mean=float(np.mean([g[k]['row_rel_mean'] for g in gathered])),
                max_abs_over_max_ref=max(g[k]['max_abs_over_max_ref'] for g in gathered),
                rows_sent_per_rank_per_batch=float(np.mean([g[k]['rows_sent_per_batch'] for g in gathered])))
        print(json.dumps(line), flush=True)
    tdist.destroy_process_group()


if __name__ == '__main__':
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    Bg = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(world, port, nb, Bg, '/tmp'), nprocs=world, join=True)
