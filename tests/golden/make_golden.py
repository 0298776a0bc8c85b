#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by running the
reference implementation (/root/reference, read-only) on small synthetic
interaction streams.

Runs ONLY in the build container (the reference never travels to the GPU box).
What it records is data: inputs, expected outputs, library versions.  Model
weights are not stored - they are regenerated from tests/golden/_weights.py.

Third-party boundary: the reference imports `torch_scatter.scatter_max`
(tiger/model/utils.py:7,15), which is not installed and not vendored.  Its
published semantics (per-segment max plus the position of that max) are restated
below as `scatter_max`; among equal maxima the FIRST position wins, which is the
torch_scatter CPU behaviour.  Everything downstream of that tie rule is
therefore "parity unpinned" by the reference itself and pinned by these vectors.

usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
from _weights import golden_param  # noqa: E402


def _install_scatter_max():
    mod = types.ModuleType('torch_scatter')

    def scatter_max(src, index, dim=-1, out=None, dim_size=None):
        n = int(index.max().item()) + 1 if dim_size is None else dim_size
        mx = torch.full((n,), float('-inf'), dtype=src.dtype).scatter_reduce(
            0, index, src, 'amax', include_self=True)
        pos = torch.arange(len(src))
        cand = torch.where(src == mx[index], pos, torch.full_like(pos, len(src)))
        arg = torch.full((n,), len(src), dtype=torch.long).scatter_reduce(
            0, index, cand, 'amin', include_self=True)
        return mx, arg

    mod.scatter_max = scatter_max
    sys.modules['torch_scatter'] = mod


_install_scatter_max()
sys.path.insert(0, '/root/reference')
from tiger.data.data_loader import GraphCollator, InteractionData  # noqa: E402
from tiger.data.graph import Graph  # noqa: E402
from tiger.model.feature_getter import NumericalFeature  # noqa: E402
from tiger.model.restarters import SeqRestarter, StaticRestarter  # noqa: E402
from tiger.model.tiger import TIGER  # noqa: E402
from tiger.model.utils import anonymized_reindex, select_latest_nids  # noqa: E402

VERSIONS = np.array([f'torch={torch.__version__}', f'numpy={np.__version__}',
                     'reference=yzhang1918/www2023tiger@v1.0.1',
                     'scatter_max=stand-in(first-index-wins)'])


def make_stream(seed, n_u, n_i, E, T, integer_ts=True):
    """Bipartite stream, ids: 0 = padding, users 1..n_u, items n_u+1..n_u+n_i."""
    rs = np.random.RandomState(seed)
    pu = 1.0 / np.arange(1, n_u + 1) ** 0.8
    pi = 1.0 / np.arange(1, n_i + 1) ** 1.0
    src = rs.choice(n_u, E, p=pu / pu.sum()) + 1
    dst = rs.choice(n_i, E, p=pi / pi.sum()) + 1 + n_u
    ts = np.sort(rs.uniform(0, T, E))
    if integer_ts:
        ts = np.floor(ts)
    eids = np.arange(1, E + 1)
    return src.astype(np.int64), dst.astype(np.int64), ts.astype(np.float64), eids.astype(np.int64)


# --------------------------------------------------------------------------- sampler
def gen_sampler():
    out = {'versions': VERSIONS}
    src, dst, ts, eids = make_stream(11, 30, 12, 1500, 400.0)  # many duplicate ts
    # a few self-describing corner cases appended: a node with one event, id gaps
    src = np.concatenate([src, [45, 45]])
    dst = np.concatenate([dst, [47, 45]])  # (45,45): self loop
    ts = np.concatenate([ts, [401.0, 402.0]])
    eids = np.concatenate([eids, [1501, 1502]])
    labels = np.zeros(len(src), dtype=np.int64)
    data = InteractionData(src, dst, ts, eids, labels, seed=0, eval=True)
    out.update(src=src, dst=dst, ts=ts, eids=eids)
    rs = np.random.RandomState(5)
    n_nodes = max(src.max(), dst.max()) + 1
    Q = 400
    q_nids = rs.randint(0, n_nodes, Q).astype(np.int64)
    q_ts = rs.choice(np.concatenate([ts, ts + 0.5, [0.0, -1.0, 1e9]]), Q).astype(np.float64)
    # force exact-hit queries (strict '<'), empty histories, over-full histories
    q_nids[:40] = src[rs.randint(0, len(src), 40)]
    q_ts[:40] = ts[rs.randint(0, len(ts), 40)]
    q_nids[40:50] = 0
    q_nids[50:60] = 46  # node id that never occurs
    out.update(q_nids=q_nids, q_ts=q_ts)
    for strategy in ('recent_edges', 'recent_nodes'):
        for K in (1, 5, 10, 40):
            g = Graph.from_data(data, strategy=strategy, seed=3)
            res = g.sample_temporal_neighbor(q_nids, q_ts, K)
            for nm, a in zip(('nbr', 'eid', 'ts', 'dir'), res):
                out[f'{strategy}_K{K}_{nm}'] = a
    # uniform consumes the graph's RandomState per non-empty query, in query order
    for K in (5, 10):
        g = Graph.from_data(data, strategy='uniform', seed=3)
        for rep in range(2):  # second call continues the same stream
            res = g.sample_temporal_neighbor(q_nids, q_ts, K)
            for nm, a in zip(('nbr', 'eid', 'ts', 'dir'), res):
                out[f'uniform_K{K}_rep{rep}_{nm}'] = a
    g = Graph.from_data(data, strategy='recent_edges', seed=3)
    out['num_node'] = np.int64(g.num_node)
    # float32-rounded query timestamps (restart path, restarters.py:70)
    q32 = q_ts.astype(np.float32)
    res = g.get_history(q_nids, q32, 8)
    for nm, a in zip(('nbr', 'eid', 'ts', 'dir'), res):
        out[f'hist32_H8_{nm}'] = a
    # select_latest_nids with ties; both float32 and float64 timestamps
    for i, (n, dt) in enumerate(((64, np.float32), (300, np.float64), (7, np.float32))):
        ids = rs.randint(0, 20, n).astype(np.int64)
        t = np.floor(rs.uniform(0, 6, n)).astype(dt)
        u, idx = select_latest_nids(torch.from_numpy(ids), torch.from_numpy(t))
        out[f'sel{i}_ids'] = ids
        out[f'sel{i}_ts'] = t
        out[f'sel{i}_unique'] = u.numpy()
        out[f'sel{i}_index'] = idx.numpy()
    # anonymized_reindex on real histories and on a hand-made block
    hist = g.get_history(q_nids, q_ts, 12)[0]
    out['anon_in'] = hist
    out['anon_out'] = anonymized_reindex(hist)
    hand = np.array([[0, 0, 0, 0], [0, 0, 0, 9], [3, 3, 3, 3], [1, 2, 1, 2], [0, 5, 6, 5], [4, 3, 2, 1]])
    out['anon2_in'] = hand
    out['anon2_out'] = anonymized_reindex(hand)
    np.savez_compressed(os.path.join(HERE, 'sampler.npz'), **out)
    print('sampler.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')


# --------------------------------------------------------------------------- model
def build_reference_model(cfg, nfeats, efeats, graph, n_edges, dropout=0.1):
    d = cfg['d']
    fg = NumericalFeature(None if nfeats is None else torch.from_numpy(nfeats).float(),
                          None if efeats is None else torch.from_numpy(efeats).float(),
                          dim=d, register_buffer=True, device=torch.device('cpu'))
    fg.n_nodes = graph.num_node
    fg.n_edges = n_edges
    if cfg['restarter'] == 'seq':
        rst = SeqRestarter(raw_feat_getter=fg, graph=graph, hist_len=cfg['H'], n_head=2, dropout=dropout)
    else:
        rst = StaticRestarter(raw_feat_getter=fg, graph=graph)
    model = TIGER(raw_feat_getter=fg, graph=graph, restarter=rst, n_neighbors=cfg['K'],
                  hit_type=cfg.get('hit', 'bin'), n_layers=cfg.get('L', 1), n_head=2, dropout=dropout,
                  msg_src=cfg['msg_src'], upd_src=cfg['upd_src'],
                  msg_tsfm_type=cfg.get('tsfm', 'id'), mem_update_type=cfg.get('upd_fn', 'gru'),
                  tgn_mode=True, msg_last_only=True)
    names, shapes = [], []
    with torch.no_grad():
        for name, p in model.named_parameters():  # de-duplicated by identity
            p.copy_(torch.from_numpy(golden_param(name, p.shape, cfg['wseed'])))
            names.append(name)
            shapes.append(list(p.shape))
    model.eval()
    return model, names, shapes


def snapshot(model, out, tag):
    out[f'{tag}_left_vals'] = model.left_memory.vals.numpy().copy()
    out[f'{tag}_left_ts'] = model.left_memory.update_ts.numpy().copy()
    out[f'{tag}_right_vals'] = model.right_memory.vals.numpy().copy()
    out[f'{tag}_right_ts'] = model.right_memory.update_ts.numpy().copy()
    has = np.array(sorted(int(x) for x in model.msg_store.nodes_with_messages), dtype=np.int64)
    out[f'{tag}_has_msg'] = has
    out[f'{tag}_msg_vals'] = model.msg_store.node_msg_vals.numpy()[has].copy()
    out[f'{tag}_msg_ts'] = model.msg_store.node_msg_ts.numpy()[has].copy()


def gen_model(name, cfg):
    d = cfg['d']
    src, dst, ts, eids = make_stream(cfg['seed'], cfg['n_u'], cfg['n_i'], cfg['E'], cfg['T'],
                                     integer_ts=cfg.get('integer_ts', True))
    E = len(src)
    n_nodes = int(max(src.max(), dst.max())) + 1
    rs = np.random.RandomState(cfg['seed'] + 100)
    nfeats = efeats = None
    if cfg.get('nfeat', 'rand') == 'rand':
        nfeats = rs.standard_normal((n_nodes, d)).astype(np.float32) * 0.5
        nfeats[0] = 0
    elif cfg.get('nfeat') == 'zero':
        nfeats = np.zeros((n_nodes, d), dtype=np.float32)
    if cfg.get('efeat', 'rand') == 'rand':
        efeats = rs.standard_normal((E + 1, cfg.get('d_e', d))).astype(np.float32)
        efeats[0] = 0
    labels = np.zeros(E, dtype=np.int64)
    neg = rs.randint(cfg['n_u'] + 1, n_nodes, E).astype(np.int64)
    data = InteractionData(src, dst, ts, eids, labels, seed=0, eval=True, neg_dst=neg)
    graph = Graph.from_data(data, strategy='recent_edges', seed=0)
    model, pnames, pshapes = build_reference_model(cfg, nfeats, efeats, graph, E)
    L = cfg.get('L', 1)  # embedding layers: layers[L] are the neighbours of the batch nodes, layers[1] the deepest hop
    collator = GraphCollator(graph, cfg['K'], L, restarter=cfg['restarter'], hist_len=cfg.get('H'))

    out = {'versions': VERSIONS, 'src': src, 'dst': dst, 'ts': ts, 'eids': eids, 'neg': neg,
           'n_nodes': np.int64(n_nodes), 'param_names': np.array(pnames),
           'param_shapes': np.array([','.join(map(str, s)) for s in pshapes]),
           'cfg': np.array([f'{k}={v}' for k, v in sorted(cfg.items())])}
    if nfeats is not None and cfg.get('nfeat') != 'zero':
        out['nfeats'] = nfeats
    if efeats is not None:
        out['efeats'] = efeats

    B = cfg['B']
    n_batches = cfg['n_batches']
    restart_at = cfg.get('restart_at', -1)
    restarting = False
    uptodate = set()
    with torch.no_grad():
        for b in range(n_batches):
            lo, hi = b * B, min((b + 1) * B, E)
            batch = [data[i] for i in range(lo, hi)]
            s, dd, ng, t, ee, _, cg = collator(batch)
            tag = f'b{b}'
            # ---- collator outputs
            out[f'{tag}_l1_nids'] = cg.layers[L][0].numpy().copy()
            out[f'{tag}_l1_eids'] = cg.layers[L][1].numpy().copy()
            out[f'{tag}_l1_ts'] = cg.layers[L][2].numpy().copy()
            for depth in range(1, L):  # deeper hops, sampled at the neighbours' timestamps (data_loader.py:131)
                out[f'{tag}_hop{L - depth + 1}_nids'] = cg.layers[depth][0].numpy().copy()
                out[f'{tag}_hop{L - depth + 1}_eids'] = cg.layers[depth][1].numpy().copy()
                out[f'{tag}_hop{L - depth + 1}_ts'] = cg.layers[depth][2].numpy().copy()
            out[f'{tag}_involved'] = cg.np_computation_graph_nodes.copy()
            rd = cg.restart_data
            out[f'{tag}_rd_index'] = rd.index.numpy().copy()
            out[f'{tag}_rd_nids'] = rd.nids.numpy().copy()
            out[f'{tag}_rd_ts'] = rd.ts.numpy().copy()
            if cfg['restarter'] == 'seq':
                out[f'{tag}_rd_hist_nids'] = rd.hist_nids.numpy().copy()
                out[f'{tag}_rd_anon'] = rd.anonymized_ids.numpy().copy()
                out[f'{tag}_rd_hist_eids'] = rd.hist_eids.numpy().copy()
                out[f'{tag}_rd_hist_ts'] = rd.hist_ts.numpy().copy()
                out[f'{tag}_rd_hist_dirs'] = rd.hist_dirs.numpy().copy()
            else:
                out[f'{tag}_rd_prev_ts'] = rd.prev_ts.numpy().copy()
            for nm, h in zip(('src_hits', 'dst_hits', 'neg_src_hits', 'neg_dst_hits'), cg.hit_data):
                out[f'{tag}_{nm}'] = h.numpy().copy()
            # ---- lazy restart exactly as train_self_supervised.py:152-163
            if b == restart_at:
                restarting = True
                uptodate = set()
                model.msg_store.clear()
            if restarting:
                involved = cg.np_computation_graph_nodes
                r_nodes = np.array(sorted(set(involved.tolist()) - uptodate), dtype=np.int64)
                r_nids = torch.from_numpy(r_nodes).long()
                r_ts = torch.full((len(r_nids),), t.min().item())
                out[f'{tag}_restart_nids'] = r_nodes
                out[f'{tag}_restart_ts'] = r_ts.numpy().copy()
                if len(r_nids):
                    hl, hr, pt = model.restarter_fn(r_nids, r_ts)
                    out[f'{tag}_restart_h_left'] = hl.numpy().copy()
                    out[f'{tag}_restart_h_right'] = hr.numpy().copy()
                    out[f'{tag}_restart_prev_ts'] = pt.numpy().copy()
                model.restart(r_nids, r_ts)
                uptodate.update(r_nodes.tolist())
                snapshot(model, out, f'{tag}_afterrestart')
            # ---- the batch itself (tiger.py:174-290)
            loss, h_left, pos, negs, hpl, hpr = model.contrast_learning(s, dd, ng, t, ee, cg)
            out[f'{tag}_loss'] = np.float32(loss.item())
            out[f'{tag}_h_left'] = h_left.numpy().copy()
            out[f'{tag}_pos_scores'] = pos.numpy().copy()
            out[f'{tag}_neg_scores'] = negs.numpy().copy()
            out[f'{tag}_h_prev_left'] = hpl.numpy().copy()
            out[f'{tag}_h_prev_right'] = hpr.numpy().copy()
            # ---- mutual-learning surrogate (tiger.py:576-590)
            index = cg.restart_data.index
            u_nids = torch.cat([s, dd])[index]
            u_ts = t.repeat(2)[index]
            sl, sr, spt = model.restarter_fn(u_nids, u_ts, cg)
            out[f'{tag}_sur_left'] = sl.numpy().copy()
            out[f'{tag}_sur_right'] = sr.numpy().copy()
            out[f'{tag}_sur_prev_ts'] = spt.numpy().copy()
            targets = torch.cat([hpl[index], hpr[index]], 0)
            preds = torch.cat([sl, sr], 0)
            valid = torch.where(~(targets == 0).all(1))[0]
            ml = model.mutual_loss_fn(preds[valid], targets[valid]).item() if len(valid) else 0.0
            out[f'{tag}_mutual_loss'] = np.float32(ml)
            if b < cfg.get('state_batches', n_batches):
                snapshot(model, out, tag)
        # flush_msg (tiger.py:444-455): consume everything that is pending
        model.flush_msg()
        snapshot(model, out, 'flushed')
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    print(f'{name}.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')


def gen_train(name, cfg):
    """The training loop of train_self_supervised.py:143-171 (dropout 0, Adam) for a few batches:
    per-batch losses, the gradients of every parameter at selected batches, final parameters
    and memories."""
    d = cfg['d']
    src, dst, ts, eids = make_stream(cfg['seed'], cfg['n_u'], cfg['n_i'], cfg['E'], cfg['T'],
                                     integer_ts=cfg.get('integer_ts', True))
    E = len(src)
    n_nodes = int(max(src.max(), dst.max())) + 1
    rs = np.random.RandomState(cfg['seed'] + 100)
    nfeats = rs.standard_normal((n_nodes, d)).astype(np.float32) * 0.5
    nfeats[0] = 0
    if cfg.get('nfeat') == 'zero':  # the JODIE sets: an all-zero node-feature table (feature_getter.py:25-47)
        nfeats = np.zeros((n_nodes, d), dtype=np.float32)
    efeats = rs.standard_normal((E + 1, cfg.get('d_e', d))).astype(np.float32)
    efeats[0] = 0
    labels = np.zeros(E, dtype=np.int64)
    neg = rs.randint(cfg['n_u'] + 1, n_nodes, E).astype(np.int64)
    data = InteractionData(src, dst, ts, eids, labels, seed=0, eval=True, neg_dst=neg)
    graph = Graph.from_data(data, strategy='recent_edges', seed=0)
    model, pnames, pshapes = build_reference_model(cfg, nfeats, efeats, graph, E, dropout=0.0)
    collator = GraphCollator(graph, cfg['K'], cfg.get('L', 1), restarter=cfg['restarter'], hist_len=cfg.get('H'))
    out = {'versions': VERSIONS, 'src': src, 'dst': dst, 'ts': ts, 'eids': eids, 'neg': neg,
           'n_nodes': np.int64(n_nodes), 'param_names': np.array(pnames),
           'param_shapes': np.array([','.join(map(str, s)) for s in pshapes]),
           'cfg': np.array([f'{k}={v}' for k, v in sorted(cfg.items())]),
           'nfeats': nfeats, 'efeats': efeats}
    B, n_batches = cfg['B'], cfg['n_batches']
    contrast_only = bool(cfg.get('contrast_only', 0))
    coef = float(cfg['mutual_coef'])
    optimizer = torch.optim.Adam(model.parameters(), lr=cfg['lr'])
    model.train()
    model.reset()
    restarting, uptodate = False, set()
    for b in range(n_batches):
        lo, hi = b * B, min((b + 1) * B, E)
        s, dd, ng, t, ee, _, cg = collator([data[i] for i in range(lo, hi)])
        s, dd, ng, ee, t = s.long(), dd.long(), ng.long(), ee.long(), t.float()
        optimizer.zero_grad()
        if b == cfg.get('restart_at', -1):
            restarting, uptodate = True, set()
            model.msg_store.clear()
        if restarting:
            involved = cg.np_computation_graph_nodes
            r_nodes = np.array(sorted(set(involved.tolist()) - uptodate), dtype=np.int64)
            r_nids = torch.from_numpy(r_nodes).long()
            model.restart(r_nids, torch.full((len(r_nids),), t.min().item()))
            uptodate.update(r_nodes.tolist())
        c_loss, m_loss = model.contrast_and_mutual_learning(s, dd, ng, t, ee, cg, contrast_only=contrast_only)
        loss = c_loss + coef * m_loss
        loss.backward()
        out[f'b{b}_contrast_loss'] = np.float32(c_loss.item())
        out[f'b{b}_mutual_loss'] = np.float32(m_loss.item())
        if b in cfg['grad_batches']:
            for nm, p_ in model.named_parameters():
                out[f'b{b}_grad.{nm}'] = (torch.zeros_like(p_) if p_.grad is None else p_.grad).numpy().copy()
        optimizer.step()
    for nm, p_ in model.named_parameters():
        out[f'final.{nm}'] = p_.detach().numpy().copy()
    with torch.no_grad():
        snapshot(model, out, 'final')
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    print(f'{name}.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')


def gen_checkpoint(name, cfg):
    """train_self_supervised.py:208-209,110-114: `model.flush_msg(); torch.save(model.state_dict(), ...)` and,
    in a later process, `model.load_state_dict(torch.load(...))`.  Stream `ckpt_at` batches, flush, take the
    reference's state_dict (every key and tensor, alias keys included), load it into a FRESH reference model built
    with other weights, and continue the stream: the outputs of the following batches are what a build that loads
    this checkpoint must reproduce."""
    d = cfg['d']
    src, dst, ts, eids = make_stream(cfg['seed'], cfg['n_u'], cfg['n_i'], cfg['E'], cfg['T'])
    E = len(src)
    n_nodes = int(max(src.max(), dst.max())) + 1
    rs = np.random.RandomState(cfg['seed'] + 100)
    nfeats = rs.standard_normal((n_nodes, d)).astype(np.float32) * 0.5
    nfeats[0] = 0
    efeats = rs.standard_normal((E + 1, d)).astype(np.float32)
    efeats[0] = 0
    labels = np.zeros(E, dtype=np.int64)
    neg = rs.randint(cfg['n_u'] + 1, n_nodes, E).astype(np.int64)
    data = InteractionData(src, dst, ts, eids, labels, seed=0, eval=True, neg_dst=neg)
    graph = Graph.from_data(data, strategy='recent_edges', seed=0)
    model, pnames, pshapes = build_reference_model(cfg, nfeats, efeats, graph, E)
    collator = GraphCollator(graph, cfg['K'], 1, restarter=cfg['restarter'], hist_len=cfg.get('H'))
    out = {'versions': VERSIONS, 'src': src, 'dst': dst, 'ts': ts, 'eids': eids, 'neg': neg,
           'n_nodes': np.int64(n_nodes), 'nfeats': nfeats, 'efeats': efeats,
           'cfg': np.array([f'{k}={v}' for k, v in sorted(cfg.items())])}
    B, at = cfg['B'], cfg['ckpt_at']
    batch = lambda b: collator([data[i] for i in range(b * B, (b + 1) * B)])
    with torch.no_grad():
        for b in range(at):
            s, dd, ng, t, ee, _, cg = batch(b)
            model.contrast_learning(s, dd, ng, t, ee, cg)
        model.flush_msg()
        sd = model.state_dict()
        out['sd_keys'] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out['sd.' + k] = v.numpy().copy()
        other = dict(cfg, wseed=cfg['wseed'] + 1000)
        fresh, _, _ = build_reference_model(other, nfeats, efeats, graph, E)
        res = fresh.load_state_dict(sd)  # strict
        assert not res.missing_keys and not res.unexpected_keys
        for b in range(at, at + cfg['n_after']):
            s, dd, ng, t, ee, _, cg = batch(b)
            loss, h_left, pos, negs, _, _ = fresh.contrast_learning(s, dd, ng, t, ee, cg)
            out[f'b{b}_loss'] = np.float32(loss.item())
            out[f'b{b}_h_left'] = h_left.numpy().copy()
            out[f'b{b}_pos_scores'] = pos.numpy().copy()
            out[f'b{b}_neg_scores'] = negs.numpy().copy()
        snapshot(fresh, out, 'final')
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    print(f'{name}.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw', len(sd), 'state_dict keys')


def gen_eval(name, cfg):
    """tiger/eval_utils.py: warmup + eval_edge_prediction over DataLoaders (AP / AUC per
    `mean_over_n_samples` events), in restart mode and in plain streaming mode."""
    from torch.utils.data import DataLoader
    from tiger.eval_utils import eval_edge_prediction, warmup
    d = cfg['d']
    src, dst, ts, eids = make_stream(cfg['seed'], cfg['n_u'], cfg['n_i'], cfg['E'], cfg['T'])
    E = len(src)
    n_nodes = int(max(src.max(), dst.max())) + 1
    rs = np.random.RandomState(cfg['seed'] + 100)
    nfeats = rs.standard_normal((n_nodes, d)).astype(np.float32) * 0.5
    nfeats[0] = 0
    efeats = rs.standard_normal((E + 1, d)).astype(np.float32)
    efeats[0] = 0
    labels = np.zeros(E, dtype=np.int64)
    data = InteractionData(src, dst, ts, eids, labels, seed=0, eval=True)  # pre-sampled negatives (bs=200)
    graph = Graph.from_data(data, strategy='recent_edges', seed=0)
    model, pnames, pshapes = build_reference_model(cfg, nfeats, efeats, graph, E, dropout=0.0)
    collator = GraphCollator(graph, cfg['K'], 1, restarter=cfg['restarter'], hist_len=cfg.get('H'))
    out = {'versions': VERSIONS, 'src': src, 'dst': dst, 'ts': ts, 'eids': eids, 'neg': data.neg_dst,
           'n_nodes': np.int64(n_nodes), 'param_names': np.array(pnames),
           'param_shapes': np.array([','.join(map(str, s)) for s in pshapes]),
           'cfg': np.array([f'{k}={v}' for k, v in sorted(cfg.items())]), 'nfeats': nfeats, 'efeats': efeats}
    B = cfg['B']
    dev = torch.device('cpu')
    mk = lambda lo, hi: DataLoader(data.get_subset(lo, hi), batch_size=B, shuffle=False, collate_fn=collator)
    n_warm, n_val = cfg['n_warm'], cfg['n_val']
    # plain streaming evaluation from an empty state
    model.reset()
    ap, auc = eval_edge_prediction(model, mk(0, n_val), dev, restart_mode=False, mean_over_n_samples=cfg['chunk'])
    out['stream_ap'], out['stream_auc'] = np.float64(ap), np.float64(auc)
    # restart mode: warm-up on the first block, evaluate the next one (train_self_supervised.py:179-202)
    model.reset()
    up = warmup(model, mk(0, n_warm), dev)
    out['warm_uptodate'] = np.array(sorted(int(x) for x in up), dtype=np.int64)
    state = model.save_memory_state()
    ap, auc = eval_edge_prediction(model, mk(n_warm, n_warm + n_val), dev, restart_mode=True,
                                   uptodate_nodes=set(up), mean_over_n_samples=cfg['chunk'])
    out['restart_ap'], out['restart_auc'] = np.float64(ap), np.float64(auc)
    model.load_memory_state(state)  # rewind and evaluate with the default 200-event windows
    ap, auc = eval_edge_prediction(model, mk(n_warm, n_warm + n_val), dev, restart_mode=True, uptodate_nodes=set(up))
    out['restart200_ap'], out['restart200_auc'] = np.float64(ap), np.float64(auc)
    with torch.no_grad():
        snapshot(model, out, 'final')
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    print(f'{name}.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')


def write_jodie_files(root, name, src, dst, ts, labels, efeats, nfeats):
    """The preprocessed JODIE layout the reference reads: data/ml_<name>.csv (+ .npy, _node.npy)."""
    import pandas as pd
    os.makedirs(os.path.join(root, 'data'), exist_ok=True)
    df = pd.DataFrame({'u': src, 'i': dst, 'ts': ts, 'label': labels, 'idx': np.arange(1, len(src) + 1)})
    df.to_csv(os.path.join(root, 'data', f'ml_{name}.csv'))
    if efeats is not None:
        np.save(os.path.join(root, 'data', f'ml_{name}.npy'), efeats)
    if nfeats is not None:
        np.save(os.path.join(root, 'data', f'ml_{name}_node.npy'), nfeats)


def gen_input_side():
    """load_jodie_data splits (data_loader.py:316-404) on a toy dataset and ChunkSampler ranges."""
    import tempfile
    from tiger.data.data_loader import ChunkSampler, load_jodie_data
    out = {'versions': VERSIONS}
    src, dst, ts, _ = make_stream(41, 120, 40, 3000, 5000.0, integer_ts=False)
    rs = np.random.RandomState(41)
    labels = (rs.uniform(size=len(src)) < 0.02).astype(np.int64)
    efeats = rs.standard_normal((len(src) + 1, 6)).astype(np.float32)
    nfeats = np.zeros((int(max(src.max(), dst.max())) + 1, 6), dtype=np.float32)
    out.update(src=src, dst=dst, ts=ts, labels=labels, efeats=efeats, nfeats=nfeats)
    with tempfile.TemporaryDirectory() as root:
        write_jodie_files(root, 'toy', src, dst, ts, labels, efeats, nfeats)
        for seed in (0, 7):
            res = load_jodie_data('toy', train_seed=seed, root=root)
            names = ('full', 'train', 'val', 'test', 'ind_val', 'ind_test')
            for nm, dset in zip(names, res[2:]):
                out[f's{seed}_{nm}_eids'] = np.asarray(dset.eids)
                if dset.neg_dst is not None:
                    out[f's{seed}_{nm}_neg'] = np.asarray(dset.neg_dst)
            # the training split draws negatives on the fly from its own sampler
            tr = res[3]
            out[f's{seed}_train_draws'] = np.array([tr[i][2] for i in range(50)])
        write_jodie_files(root, 'bare', src, dst, ts, labels, None, None)  # no feature files
        res = load_jodie_data('bare', train_seed=1, root=root, val_p=0.6, test_p=0.8)
        assert res[0] is None and res[1] is None
        out['bare_train_eids'] = np.asarray(res[3].eids)
        out['bare_ind_test_eids'] = np.asarray(res[7].eids)
    rows = []
    for (n, rank, ws, bs, seed, epoch) in [(1000, 0, 2, 64, 0, 0), (1000, 1, 2, 64, 0, 0), (1000, 1, 2, 64, 0, 3),
                                           (12345, 2, 4, 200, 5, 1), (512, 0, 1, 64, 9, 2), (130, 1, 2, 64, 1, 4)]:
        cs = ChunkSampler(n, rank, ws, bs, seed)
        cs.set_epoch(epoch)
        idx = list(iter(cs))
        rows.append([n, rank, ws, bs, seed, epoch, len(cs), idx[0] if idx else -1, idx[-1] if idx else -1])
    out['chunk_sampler'] = np.array(rows, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'input_side.npz'), **out)
    print('input_side.npz', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')


EVAL_SCENARIOS = {
    'eval_seq_lr_d8': dict(d=8, n_u=40, n_i=15, E=900, T=450.0, B=50, K=5, H=8, seed=31, wseed=31, restarter='seq',
                           msg_src='left', upd_src='right', hit='bin', n_warm=300, n_val=500, chunk=64),
    'eval_static_ll_d16': dict(d=16, n_u=60, n_i=25, E=900, T=500.0, B=64, K=10, seed=32, wseed=32, restarter='static',
                               msg_src='left', upd_src='left', hit='vec', n_warm=256, n_val=600, chunk=100),
}

TRAIN_SCENARIOS = {
    # CLI defaults: seq restarter, msg=left upd=right, 'bin' hits, mutual learning with a lazy restart
    'train_seq_lr_d8': dict(d=8, n_u=40, n_i=15, E=480, T=300.0, B=40, n_batches=10, K=5, H=8, seed=21, wseed=21,
                            restarter='seq', msg_src='left', upd_src='right', hit='bin', restart_at=6,
                            lr=1e-2, mutual_coef=1.0, grad_batches=(0, 1, 4, 7)),
    # the same recipe on an all-zero node-feature table (every JODIE set), longer histories than most nodes have: the
    # restarter's narrow form (csrc/tg_restart.hip: no node-feature blocks, tabulated anony_emb block, shared padded rows)
    'train_seq_lr_d8_zeronf': dict(d=8, n_u=40, n_i=15, E=480, T=300.0, B=40, n_batches=9, K=5, H=12, seed=28, wseed=28,
                                   nfeat='zero', restarter='seq', msg_src='left', upd_src='right', hit='bin', restart_at=5,
                                   lr=1e-2, mutual_coef=1.0, grad_batches=(0, 1, 4, 7)),
    # C2-shaped: msg=left upd=left, static restarter, 'vec' hits, wider edge features
    'train_static_ll_d16': dict(d=16, d_e=12, n_u=60, n_i=25, E=640, T=500.0, B=64, n_batches=8, K=10, seed=22,
                                wseed=22, restarter='static', msg_src='left', upd_src='left', hit='vec',
                                restart_at=5, lr=1e-2, mutual_coef=0.5, grad_batches=(1, 3, 6)),
    # restart_prob == 0: contrast loss only (tiger.py:570-572), TGN-style right/right
    # non-default message transforms / updater (--tsfm_fn, --upd_fn)
    'train_mlp_merge_d8': dict(d=8, n_u=30, n_i=12, E=300, T=200.0, B=30, n_batches=8, K=6, H=6, seed=24, wseed=24,
                               restarter='seq', msg_src='left', upd_src='right', tsfm='mlp', upd_fn='merge', hit='bin',
                               restart_at=5, lr=1e-2, mutual_coef=1.0, grad_batches=(1, 3, 6)),
    'train_linear_gru_d8': dict(d=8, n_u=30, n_i=12, E=300, T=200.0, B=30, n_batches=8, K=6, seed=25, wseed=25,
                                restarter='static', msg_src='right', upd_src='left', tsfm='linear', hit='count',
                                lr=1e-2, mutual_coef=1.0, grad_batches=(1, 4)),
    'train_contrast_rr_d8': dict(d=8, n_u=30, n_i=30, E=400, T=500.0, B=50, n_batches=6, K=10, H=6, seed=23,
                                 wseed=23, integer_ts=False, restarter='seq', msg_src='right', upd_src='right',
                                 hit='none', contrast_only=1, lr=1e-2, mutual_coef=1.0, grad_batches=(1, 4)),
    # --n_layers 2 (the constructor default, tiger.py:29): gradients through both attention layers
    'train_static_lr_d8_L2': dict(d=8, n_u=30, n_i=12, E=300, T=200.0, B=30, n_batches=8, K=4, L=2, seed=26, wseed=26,
                                  restarter='static', msg_src='left', upd_src='right', hit='bin', restart_at=5,
                                  lr=1e-2, mutual_coef=1.0, grad_batches=(1, 3, 6)),
    'train_contrast_ll_d16_L2': dict(d=16, n_u=40, n_i=15, E=320, T=240.0, B=40, n_batches=6, K=6, H=6, L=2, seed=27,
                                     wseed=27, restarter='seq', msg_src='left', upd_src='left', hit='vec',
                                     contrast_only=1, lr=1e-2, mutual_coef=1.0, grad_batches=(1, 4)),
}

CKPT_SCENARIOS = {
    'ckpt_seq_lr_d8': dict(d=8, n_u=40, n_i=15, E=400, T=300.0, B=40, K=5, H=8, seed=51, wseed=51, restarter='seq',
                           msg_src='left', upd_src='right', hit='bin', ckpt_at=4, n_after=3),
    'ckpt_static_ll_d16': dict(d=16, n_u=60, n_i=25, E=500, T=500.0, B=64, K=10, seed=52, wseed=52, restarter='static',
                               msg_src='left', upd_src='left', hit='vec', ckpt_at=3, n_after=3),
}

SCENARIOS = {
    # C1-like plumbing case: seq restarter, msg=left upd=right (CLI defaults)
    'seq_lr_d8': dict(d=8, n_u=40, n_i=15, E=600, T=300.0, B=40, n_batches=12, K=5, H=8, seed=1, wseed=1,
                      restarter='seq', msg_src='left', upd_src='right', restart_at=7),
    # C2-like: msg=left upd=left; static restarter as in C3; wider edge features than memory
    'static_ll_d16': dict(d=16, d_e=12, n_u=60, n_i=25, E=900, T=500.0, B=64, n_batches=10, K=10, seed=2, wseed=2,
                          restarter='static', msg_src='left', upd_src='left', restart_at=6),
    # msg=right upd=right (TGN-style), no feature tables at all (LastFM-style, C4)
    'seq_rr_d8_nofeat': dict(d=8, n_u=30, n_i=30, E=500, T=1.0e6, B=50, n_batches=8, K=10, H=12, seed=3, wseed=3,
                             integer_ts=False, nfeat=None, efeat=None,
                             restarter='seq', msg_src='right', upd_src='right', restart_at=5),
    # Wikipedia-shaped micro case: d=172, zero node features, large time deltas
    'static_ll_d172': dict(d=172, n_u=30, n_i=10, E=240, T=2.6e6, B=30, n_batches=6, K=10, seed=4, wseed=4,
                           nfeat='zero', restarter='static', msg_src='left', upd_src='left',
                           restart_at=4, state_batches=3),
    'seq_lr_d172': dict(d=172, n_u=24, n_i=8, E=160, T=2.6e6, B=20, n_batches=5, K=10, H=40, seed=5, wseed=5,
                        nfeat='zero', restarter='seq', msg_src='left', upd_src='right',
                        restart_at=3, state_batches=2),
    # non-default message transforms / updater
    'mlp_merge_d8': dict(d=8, n_u=30, n_i=12, E=300, T=200.0, B=30, n_batches=6, K=5, H=6, seed=6, wseed=6,
                         restarter='seq', msg_src='left', upd_src='right', tsfm='mlp', upd_fn='merge'),
    'linear_gru_d8': dict(d=8, n_u=30, n_i=12, E=300, T=200.0, B=30, n_batches=6, K=5, seed=7, wseed=7,
                          restarter='static', msg_src='right', upd_src='left', tsfm='linear', hit='vec'),
    # --n_layers 2: hop-2 sampled at the neighbours' timestamps, two attention layers
    'static_lr_d8_L2': dict(d=8, n_u=30, n_i=12, E=360, T=240.0, B=30, n_batches=8, K=4, L=2, seed=8, wseed=8,
                            restarter='static', msg_src='left', upd_src='right', restart_at=5),
    'seq_ll_d16_L2': dict(d=16, n_u=40, n_i=15, E=300, T=200.0, B=40, n_batches=5, K=5, H=6, L=2, seed=9, wseed=9,
                          nfeat='zero', restarter='seq', msg_src='left', upd_src='left'),
}

if __name__ == '__main__':
    only = sys.argv[1:]
    if not only or 'sampler' in only:
        gen_sampler()
    for nm, cfg in SCENARIOS.items():
        if not only or nm in only:
            gen_model(nm, cfg)
    for nm, cfg in TRAIN_SCENARIOS.items():
        if not only or nm in only:
            gen_train(nm, cfg)
    if not only or 'input_side' in only:
        gen_input_side()
    for nm, cfg in EVAL_SCENARIOS.items():
        if not only or nm in only:
            gen_eval(nm, cfg)
    for nm, cfg in CKPT_SCENARIOS.items():
        if not only or nm in only:
            gen_checkpoint(nm, cfg)
