#!/usr/bin/env python3
"""Self-supervised link prediction with TIGER on one MI355X: the recipe of the reference's
train_self_supervised.py (train with lazy restarts -> validate with memory snapshots -> checkpoint ->
test), written against this package's mirror of the reference API.

    python examples/link_prediction.py --data wikipedia --root /path/with/data/ml_wikipedia.csv ...

Only `run()` matters; the few flags exist to make the file runnable.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from www2023tiger_amd.eval_utils import eval_edge_prediction, warmup  # noqa: E402
from www2023tiger_amd.init_utils import init_data, init_model  # noqa: E402
from www2023tiger_amd.optim import Adam  # noqa: E402  (torch.optim.Adam works too; this one never stalls the loop)


def train_epoch(model, train_dl, optimizer, device, *, restart_prob, mutual_coef, rng):
    """train_self_supervised.py:143-171"""
    model.train()
    model.reset()
    losses, restarting, uptodate = [], False, set()
    for i_batch, (src, dst, neg, ts, eids, _, cg) in enumerate(train_dl):
        src, dst, neg, eids = (x.long().to(device) for x in (src, dst, neg, eids))
        ts = ts.float().to(device)
        optimizer.zero_grad()
        if rng.rand() < restart_prob and i_batch:
            restarting, uptodate = True, set()
            model.msg_store.clear()
        if restarting:  # lazy restart of the nodes this batch touches
            todo = set(cg.np_computation_graph_nodes.tolist()) - uptodate
            nids = torch.tensor(sorted(todo), dtype=torch.long, device=device)
            model.restart(nids, torch.full((len(nids),), ts.min().item(), device=device))
            uptodate |= todo
        c_loss, m_loss = model.contrast_and_mutual_learning(src, dst, neg, ts, eids, cg,
                                                            contrast_only=(restart_prob == 0))
        loss = c_loss + mutual_coef * m_loss
        loss.backward()
        optimizer.step()
        losses.append((loss.item(), c_loss.item(), float(m_loss.detach())))
    return np.array(losses)


def evaluate_pair(model, dl, ind_dl, device, restart_mode, uptodate):
    """Transductive then inductive evaluation from the same memory state (train_self_supervised.py:191-202);
    the memories end at the state after the transductive pass."""
    start = model.save_memory_state()
    ap, auc = eval_edge_prediction(model, dl, device, restart_mode, uptodate_nodes=set(uptodate))
    end = model.save_memory_state()
    model.load_memory_state(start)
    ind_ap, ind_auc = eval_edge_prediction(model, ind_dl, device, restart_mode, uptodate_nodes=set(uptodate))
    model.load_memory_state(end)
    return ap, auc, ind_ap, ind_auc


def run(data, root, *, seed=0, n_epochs=1, bs=200, lr=1e-4, dim=None, n_neighbors=10, n_heads=2, hit_type='bin',
        restarter_type='seq', hist_len=40, msg_src='left', upd_src='right', restart_prob=0.01, mutual_coef=1.0,
        warmup_steps=0, strategy='recent_edges', dropout=0.1, ckpt_path=None, device='cuda:0'):
    device = torch.device(device)
    torch.manual_seed(seed)
    rng = np.random.RandomState(seed)
    basic, (train_graph, full_graph), dls = init_data(
        data, root, seed, num_workers=0, bs=bs, warmup_steps=warmup_steps, subset=1.0, strategy=strategy, n_layers=1,
        n_neighbors=n_neighbors, restarter_type=restarter_type, hist_len=hist_len, device=device)
    nfeats, efeats, full_data = basic[:3]
    train_dl, _, val_dl, ind_val_dl, test_dl, ind_test_dl, val_warm_dl, test_warm_dl = dls
    model = init_model(nfeats, efeats, train_graph, full_graph, full_data, device, dim=dim, n_layers=1,
                       n_heads=n_heads, n_neighbors=n_neighbors, hit_type=hit_type, dropout=dropout,
                       restarter_type=restarter_type, hist_len=hist_len, msg_src=msg_src, upd_src=upd_src,
                       msg_tsfm_type='id', mem_update_type='gru')
    optimizer = Adam(model.parameters(), lr=lr)
    restart_mode = restart_prob > 0
    log = []
    for epoch in range(n_epochs):
        model.graph = train_graph
        losses = train_epoch(model, train_dl, optimizer, device, restart_prob=restart_prob, mutual_coef=mutual_coef,
                             rng=rng)
        model.eval()
        model.flush_msg()
        model.graph = full_graph
        uptodate = set()
        if restart_mode:
            model.msg_store.clear()
            if warmup_steps:
                uptodate = warmup(model, val_warm_dl, device)
        val = evaluate_pair(model, val_dl, ind_val_dl, device, restart_mode, uptodate)
        model.flush_msg()
        if ckpt_path:
            torch.save(model.state_dict(), ckpt_path)
        log.append(dict(epoch=epoch, loss=float(losses[:, 0].mean()), contrast=float(losses[:, 1].mean()),
                        mutual=float(losses[:, 2].mean()), val_ap=val[0], val_auc=val[1], ind_val_ap=val[2],
                        ind_val_auc=val[3]))
    if ckpt_path:  # the reference reloads its best checkpoint before testing
        model.load_state_dict(torch.load(ckpt_path, map_location=device))
    model.eval()
    model.graph = full_graph
    uptodate = set()
    if restart_mode:
        model.msg_store.clear()
        if warmup_steps:
            uptodate = warmup(model, test_warm_dl, device)
    test = evaluate_pair(model, test_dl, ind_test_dl, device, restart_mode, uptodate)
    return dict(epochs=log, test_ap=test[0], test_auc=test[1], ind_test_ap=test[2], ind_test_auc=test[3]), model


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('-d', '--data', default='wikipedia')
    ap.add_argument('--root', default='.')
    ap.add_argument('--n_epochs', type=int, default=1)
    ap.add_argument('--bs', type=int, default=200)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--restarter_type', default='seq', choices=['seq', 'static'])
    ap.add_argument('--restart_prob', type=float, default=0.01)
    ap.add_argument('--dropout', type=float, default=0.1)
    a = ap.parse_args()
    out, _ = run(a.data, a.root, n_epochs=a.n_epochs, bs=a.bs, lr=a.lr, restarter_type=a.restarter_type,
                 restart_prob=a.restart_prob, dropout=a.dropout)
    print(out)
