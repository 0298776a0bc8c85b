"""CPU oracle for the TIGER event-batch hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain numpy / torch-CPU restatement of the reference algorithm
(yzhang1918/www2023tiger @ v1.0.1).  It is the checker for the HIP path: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The product package (www2023tiger_amd) never imports anything from oracle/.

Pinning: every function below is checked against golden vectors produced by
running the reference itself in the build container (tests/golden/*.npz,
generator tests/golden/make_golden.py).  One third-party boundary is NOT pinned
by the reference: `torch_scatter.scatter_max` (tiger/model/utils.py:15) is not
vendored and has no lock file; its tie rule is fixed here as first-index-wins
(torch_scatter's CPU behaviour) - "parity unpinned" at that boundary.

Each function cites the reference file:line it follows (paths relative to the
reference root).
"""
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

F32 = np.float32
# inputs at least this long take the vectorised forms of the sampler / dedup loops (identical results,
# checked in tests/test_oracle_golden.py); the full-size parity cases would otherwise spend minutes in Python
VECTORISE_FROM = 256
EMBED_CHUNK = 16384


# =============================================================================
# Temporal graph (tiger/data/graph.py)
# =============================================================================
class OracleGraph:
    """Per-node time-sorted adjacency in CSR form.

    graph.py:11-42, 226-241: every event (src, dst, t, eid) adds (dst, eid, t, 0)
    to src's list and (src, eid, t, 1) to dst's list, in event order; each list is
    then stably sorted by t.  num_node = max id + 1 (id 0 is the padding node).
    """

    def __init__(self, src, dst, ts, eids, strategy='recent_edges', seed=None, max_node_id=None):
        src = np.asarray(src, dtype=np.int64)
        dst = np.asarray(dst, dtype=np.int64)
        ts = np.asarray(ts, dtype=np.float64)
        eids = np.asarray(eids, dtype=np.int64)
        if max_node_id is None:
            max_node_id = int(max(src.max(), dst.max()))
        self.num_node = max_node_id + 1
        self.strategy = strategy
        self.rng = np.random.RandomState(seed)  # graph.py:22
        E = len(src)
        owner = np.empty(2 * E, dtype=np.int64)
        owner[0::2], owner[1::2] = src, dst
        other = np.empty(2 * E, dtype=np.int64)
        other[0::2], other[1::2] = dst, src
        t2 = np.repeat(ts, 2)
        e2 = np.repeat(eids, 2)
        flag = np.tile(np.array([0, 1], dtype=np.int64), E)
        order = np.lexsort((np.arange(2 * E), t2, owner))  # by owner, then t, stable
        self.nbr = other[order]
        self.eid = e2[order]
        self.ts = t2[order]
        self.dir = flag[order]
        deg = np.bincount(owner, minlength=self.num_node)
        self.indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)

    # graph.py:44-53  searchsorted(..., side='left') == number of events with ts < t
    def find_before(self, nid: int, t: float) -> Tuple[int, int]:
        lo, hi = self.indptr[nid], self.indptr[nid + 1]
        return lo, lo + int(np.searchsorted(self.ts[lo:hi], t, side='left'))

    def sample_temporal_neighbor(self, nids, ts, n_neighbors=20, strategy=None):
        """graph.py:67-148.  Left padded with zeros; outputs i64,i64,f32,i64."""
        strategy = self.strategy if strategy is None else strategy
        bs, K = len(nids), n_neighbors
        assert len(nids) == len(ts)
        if strategy == 'recent_edges' and bs >= VECTORISE_FROM:
            return self._sample_recent_edges_vectorised(np.asarray(nids), np.asarray(ts), K)
        o_n = np.zeros((bs, K), dtype=np.int64)
        o_e = np.zeros((bs, K), dtype=np.int64)
        o_t = np.zeros((bs, K), dtype=np.float32)
        o_d = np.zeros((bs, K), dtype=np.int64)
        for i, (nid, t) in enumerate(zip(nids, ts)):
            lo, hi = self.find_before(int(nid), t)
            if hi == lo:
                continue
            if strategy == 'uniform':  # graph.py:101-115 (alpha == 0 branch)
                idx = self.rng.randint(0, hi - lo, K)
                sel = lo + idx
                sel = sel[self.ts[sel].argsort()]
            elif strategy == 'recent_edges':  # graph.py:117-127
                sel = np.arange(max(lo, hi - K), hi)
            elif strategy == 'recent_nodes':  # graph.py:129-143
                seg = self.nbr[lo:hi]
                _, first_from_end = np.unique(seg[::-1], return_index=True)
                last_pos = (hi - lo) - 1 - np.sort(first_from_end)[::-1]
                sel = lo + last_pos[-K:]
            else:
                raise NotImplementedError(strategy)
            n = len(sel)
            o_n[i, K - n:] = self.nbr[sel]
            o_e[i, K - n:] = self.eid[sel]
            o_t[i, K - n:] = self.ts[sel]
            o_d[i, K - n:] = self.dir[sel]
        return o_n, o_e, o_t, o_d

    def _sample_recent_edges_vectorised(self, nids, ts, K):
        """The 'recent_edges' branch of the loop above for all queries at once (same arithmetic: a
        'left' binary search on the float64 run of each node, then the K-entry tail, left padded).
        Only a speed-up for the full-size parity cases; tests/test_oracle_golden.py checks it equal to
        the per-query loop and against the reference's sampler vectors."""
        lo0 = self.indptr[nids]
        lo, hi = lo0.copy(), self.indptr[nids + 1].copy()
        last = max(len(self.ts) - 1, 0)
        while True:
            act = lo < hi
            if not act.any():
                break
            mid = (lo + hi) >> 1
            go = act & (self.ts[np.minimum(mid, last)] < ts)  # entries with ts < t lie left of the cut
            lo = np.where(go, mid + 1, lo)
            hi = np.where(act & ~go, mid, hi)
        idx = lo[:, None] - K + np.arange(K)[None, :]
        ok = idx >= lo0[:, None]
        idx = np.where(ok, idx, 0)
        pick = lambda arr, dt: np.where(ok, arr[idx], 0).astype(dt) if len(arr) else np.zeros(idx.shape, dtype=dt)
        return pick(self.nbr, np.int64), pick(self.eid, np.int64), pick(self.ts, np.float32), pick(self.dir, np.int64)

    def get_history(self, nids, ts, hist_len):  # graph.py:150-155
        return self.sample_temporal_neighbor(nids, ts, hist_len, strategy='recent_edges')


# =============================================================================
# tiger/model/utils.py
# =============================================================================
def select_latest_nids(nids: np.ndarray, ts: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """utils.py:10-16.  Sorted unique ids and, per id, the position of its maximum
    timestamp; among equal maxima the first position wins (scatter_max, CPU)."""
    nids = np.asarray(nids)
    ts = np.asarray(ts)
    uniq, inv = np.unique(nids, return_inverse=True)
    if len(nids) >= VECTORISE_FROM:  # same rule without the Python loop: per id the largest ts, first position among ties
        order = np.lexsort((np.arange(len(nids)), -ts, inv))
        first = np.concatenate([[True], inv[order][1:] != inv[order][:-1]])
        return uniq.astype(np.int64), order[first].astype(np.int64)
    best = np.full(len(uniq), -1, dtype=np.int64)
    for i in range(len(nids)):
        j = inv[i]
        if best[j] < 0 or ts[i] > ts[best[j]]:
            best[j] = i
    return uniq.astype(np.int64), best


def anonymized_reindex(hist_nids: np.ndarray) -> np.ndarray:
    """utils.py:19-27.  Per row, number distinct ids by first appearance scanning
    from the most recent (last) column; padding (0) stays 0."""
    out = np.zeros_like(hist_nids)
    for i, line in enumerate(hist_nids):
        od = OrderedDict.fromkeys(line[::-1].tolist())
        remap = {k: j + 1 for j, k in enumerate(od.keys())}
        out[i] = [remap[int(x)] for x in line]
    out[hist_nids == 0] = 0
    return out


# =============================================================================
# Collation (tiger/data/data_loader.py:43-168, data_classes.py:150-165)
# =============================================================================
def collate(graph: OracleGraph, src, dst, neg, ts, n_neighbors: int, restarter: str,
            hist_len: Optional[int] = None, n_layers: int = 1) -> Dict[str, np.ndarray]:
    """One batch of GraphCollator.__call__ (data_loader.py:77-93) for n_layers 1 or 2."""
    src, dst, neg = (np.asarray(x, dtype=np.int64) for x in (src, dst, neg))
    ts = np.asarray(ts, dtype=np.float64)
    nids3 = np.concatenate([src, dst, neg])
    ts3 = np.tile(ts, 3)
    l1_n, l1_e, l1_t, _ = graph.sample_temporal_neighbor(nids3, ts3, n_neighbors)  # :128
    seen = [nids3, l1_n.ravel()]
    out = {}
    if n_layers == 2:  # :131 the next hop is sampled at the NEIGHBOURS' (float32) timestamps
        h2_n, h2_e, h2_t, _ = graph.sample_temporal_neighbor(l1_n.ravel(), l1_t.ravel(), n_neighbors)
        seen.append(h2_n.ravel())
        out.update(hop2_nids=h2_n, hop2_eids=h2_e, hop2_ts=h2_t)
    elif n_layers != 1:
        raise NotImplementedError('n_layers in (1, 2)')
    involved = np.unique(np.concatenate(seen))  # :109-121 (sorted set)
    local_index = np.zeros(graph.num_node, dtype=np.int64)  # data_classes.py:163-165
    local_index[involved] = np.arange(len(involved))
    out.update(l1_nids=l1_n, l1_eids=l1_e, l1_ts=l1_t, involved=involved, local_index=local_index)
    # restart data (:133-168) on cat[src,dst], tile(ts,2) with float64 timestamps
    pos = np.concatenate([src, dst])
    ts2 = np.tile(ts, 2)
    u, idx = select_latest_nids(pos, ts2)
    tu = ts2[idx]
    out.update(rd_index=idx, rd_nids=u, rd_ts=tu.astype(np.float32))
    if restarter == 'seq':
        h_n, h_e, h_t, h_d = graph.get_history(u, tu, hist_len)
        out.update(rd_hist_nids=h_n, rd_anon=anonymized_reindex(h_n), rd_hist_eids=h_e,
                   rd_hist_ts=h_t, rd_hist_dirs=h_d)
    elif restarter == 'static':
        out['rd_prev_ts'] = graph.get_history(u, tu, 1)[2]  # [P,1] (data_loader.py:161-165)
    else:
        raise NotImplementedError(restarter)
    # hits (:61-75): neighbours always by recent_edges
    def hit(center, target):
        nb = graph.sample_temporal_neighbor(target, ts, n_neighbors, strategy='recent_edges')[0]
        return (center[:, None] == nb).astype(np.float32)
    out.update(src_hits=hit(src, dst), dst_hits=hit(dst, src),
               neg_src_hits=hit(src, neg), neg_dst_hits=hit(neg, src))
    return out


# =============================================================================
# Neural pieces, restated with explicit matmuls (float32, torch CPU)
# =============================================================================
def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def time_encode(ts: torch.Tensor, basis_freq: torch.Tensor, phase: torch.Tensor) -> torch.Tensor:
    """time_encoding.py:24-26: cos(fl32(ts * w) + phi), separate multiply and add."""
    return torch.cos(ts.unsqueeze(-1) * basis_freq + phase)


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def merge_layer(x1, x2, p, prefix, hmask=None):
    """basic_modules.py:16-19.  hmask: dropout mask (keep / (1-p)) on the hidden layer, None in eval."""
    h = torch.relu(linear(torch.cat([x1, x2], -1), p[prefix + 'fc1.weight'], p[prefix + 'fc1.bias']))
    if hmask is not None:
        h = h * hmask
    return linear(h, p[prefix + 'fc2.weight'], p[prefix + 'fc2.bias'])


def _mix32(x):
    x = x ^ (x >> np.uint64(16))
    x = (x * np.uint64(0x7feb352d)) & np.uint64(0xffffffff)
    x = x ^ (x >> np.uint64(15))
    x = (x * np.uint64(0x846ca68b)) & np.uint64(0xffffffff)
    return x ^ (x >> np.uint64(16))


def dropout_keep(seed: int, counter: int, stream: int, n: int, p: float, offset: int = 0) -> np.ndarray:
    """The library's counter-based dropout mask (csrc/tg_common.h: drop_keep): element `offset + j`
    of mask stream `stream` at step `counter` of seed `seed` is kept iff hash >= p * 2^32."""
    key = (seed + counter * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    m32 = np.uint64(0xffffffff)
    h = _mix32((idx & m32) ^ np.uint64(key & 0xffffffff))
    h = _mix32((h + (idx >> np.uint64(32)) * np.uint64(0x9e3779b9) + np.uint64(key >> 32)
                + np.uint64((stream * 0x85ebca6b) & 0xffffffff)) & m32)
    return h >= np.uint64(min(int(p * 4294967296.0), 0xffffffff))


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRUCell as used at update_modules.py:33-36."""
    d = h.shape[1]
    gi = linear(x, w_ih, b_ih)
    gh = linear(h, w_hh, b_hh)
    r = torch.sigmoid(gi[:, :d] + gh[:, :d])
    z = torch.sigmoid(gi[:, d:2 * d] + gh[:, d:2 * d])
    n = torch.tanh(gi[:, 2 * d:] + r * gh[:, 2 * d:])
    return (1 - z) * n + z * h


def mha(query, key, value, wq, wk, wv, b_in, wo, bo, n_head, key_padding_mask, drop=None):
    """torch.nn.MultiheadAttention forward (eval, batch_first=False, math path):
    query [L,n,E], key/value [S,n,*]; returns [L,n,E].  Used at
    temporal_agg_modules.py:204-227 and restarters.py:46,105."""
    L, n, E = query.shape
    S = key.shape[0]
    dh = E // n_head
    q = linear(query, wq, b_in[:E])
    k = linear(key, wk, b_in[E:2 * E])
    v = linear(value, wv, b_in[2 * E:])
    q = q.reshape(L, n * n_head, dh).transpose(0, 1) * (1.0 / np.sqrt(dh))
    k = k.reshape(S, n * n_head, dh).transpose(0, 1)
    v = v.reshape(S, n * n_head, dh).transpose(0, 1)
    att = q @ k.transpose(1, 2)  # [n*h, L, S]
    mask = key_padding_mask.view(n, 1, 1, S).expand(n, n_head, L, S).reshape(n * n_head, L, S)
    att = att.masked_fill(mask, float('-inf'))
    att = torch.softmax(att, dim=-1)
    if drop is not None:  # nn.MultiheadAttention dropout acts on the attention probabilities
        att = att * drop(att.numel()).reshape(att.shape)
    o = (att @ v).transpose(0, 1).reshape(L, n, E)
    return linear(o, wo, bo)


class OracleTIGER:
    """State + per-batch algorithm of TIGE/TIGER (tiger/model/tiger.py) for
    n_layers == 1, tgn_mode, msg_last_only - the only mode init_utils.py:166 builds.

    `params` maps reference state_dict parameter names to float32 arrays.
    """

    def __init__(self, params: Dict[str, np.ndarray], graph: OracleGraph, *, n_nodes: int, dim: int,
                 nfeats: Optional[np.ndarray], efeats: Optional[np.ndarray], n_neighbors: int,
                 msg_src: str, upd_src: str, restarter: str, hist_len: Optional[int] = None,
                 n_head: int = 2, tsfm: str = 'id', upd_fn: str = 'gru', hit_type: str = 'bin'):
        if msg_src not in ('left', 'right') or upd_src not in ('left', 'right'):  # tiger.py:156-160
            raise ValueError('invalid msg_src / upd_src')
        self.p = {k: _t(np.asarray(v, dtype=F32)) for k, v in params.items()}
        self.graph = graph
        self.N, self.d = n_nodes, dim
        self.nfeats = None if nfeats is None else _t(nfeats.astype(F32))
        self.efeats = None if efeats is None else _t(efeats.astype(F32))
        self.d_e = dim if efeats is None else efeats.shape[1]  # feature_getter.py:78
        self.K, self.H, self.n_head = n_neighbors, hist_len, n_head
        self.msg_src, self.upd_src = msg_src, upd_src
        self.restarter, self.tsfm, self.upd_fn, self.hit_type = restarter, tsfm, upd_fn, hit_type
        self.raw_msg_dim = 2 * dim + self.d_e + dim  # tiger.py:62
        self.dropout = None  # training only: (p, seed, step counter) of the library's mask generator
        self.reset()

    # ---- state (memory.py:12-52, 55-75) -------------------------------------
    def reset(self):  # tiger.py:457-463
        N, d = self.N, self.d
        self.left_vals = torch.zeros(N, d)
        self.left_ts = torch.zeros(N)
        self.right_vals = torch.zeros(N, d)
        self.right_ts = torch.zeros(N)
        self.msg_vals = torch.zeros(N, self.raw_msg_dim)
        self.msg_ts = torch.zeros(N)
        self.has_msg = np.zeros(N, dtype=bool)

    def _mem(self, which):
        return (self.left_vals, self.left_ts) if which == 'left' else (self.right_vals, self.right_ts)

    @staticmethod
    def _mem_set(vals, tss, ids, new_vals, new_ts, skip_check=False):  # memory.py:41-52
        if not skip_check:
            if (tss[ids] > new_ts).any():
                raise ValueError('You are not allowed to modify past memory.')
            if len(ids) != len(torch.unique(ids)):
                raise ValueError('Duplicate node ids are not allowed.')
        tss[ids] = new_ts.detach() if isinstance(new_ts, torch.Tensor) else new_ts
        vals[ids] = new_vals.detach()

    def _drop(self, stream: int, offset: int = 0):
        """mask factory for one dropout site: n -> float tensor keep / (1 - p), or None when off"""
        if self.dropout is None or not torch.is_grad_enabled():
            return None
        p, seed, counter = self.dropout
        return lambda n: _t(dropout_keep(seed, counter, stream, n, p, offset).astype(F32) / np.float32(1.0 - p))

    # ---- features (feature_getter.py:80-106) ----------------------------------
    def node_feat(self, ids):
        if self.nfeats is None:
            return torch.zeros(*ids.shape, self.d)
        return self.nfeats[ids]

    def edge_feat(self, eids):
        if self.efeats is None:
            return torch.zeros(*eids.shape, self.d_e)
        return self.efeats[eids]

    def te(self, ts):
        return time_encode(ts, self.p['time_encoder.basis_freq'], self.p['time_encoder.phase'])

    # ---- message transform (message_modules.py:20-55) -------------------------
    def msg_transform(self, raw):
        if self.tsfm == 'id':
            return raw
        if self.tsfm == 'linear':
            return linear(raw, self.p['msg_transform_fn.fn.1.weight'], self.p['msg_transform_fn.fn.1.bias'])
        h = torch.relu(linear(raw, self.p['msg_transform_fn.fn.1.weight'], self.p['msg_transform_fn.fn.1.bias']))
        return linear(h, self.p['msg_transform_fn.fn.4.weight'], self.p['msg_transform_fn.fn.4.bias'])

    def updater(self, mem, msg):  # update_modules.py:30-47
        if self.upd_fn == 'gru':
            pre = 'right_mem_updater.cell.'
            return gru_cell(msg, mem, self.p[pre + 'weight_ih'], self.p[pre + 'weight_hh'],
                            self.p[pre + 'bias_ih'], self.p[pre + 'bias_hh'])
        return merge_layer(msg, mem, self.p, 'right_mem_updater.fn.')

    # ---- STEP 1+2 (tiger.py:208-221, 292-356) --------------------------------
    def consume(self, node_ids: Optional[np.ndarray]):
        """Returns (outdated ids sorted, h(t'+) rows, message ts) for pending nodes
        among node_ids (all nodes if None)."""
        if node_ids is None:
            outdated = np.nonzero(self.has_msg)[0]
        else:
            node_ids = np.asarray(node_ids)
            outdated = node_ids[self.has_msg[node_ids]]
        outdated = np.unique(outdated)
        if len(outdated) == 0:
            return outdated, None, None
        o = _t(outdated)
        raw, mts = self.msg_vals[o], self.msg_ts[o]
        msg_mem_ts = self._mem(self.msg_src)[1][o]
        if (msg_mem_ts > mts).any():  # message_modules.py:158-159
            raise ValueError('Messages happened later than memory updating.')
        if self.msg_src == 'left' and not bool((mts == msg_mem_ts).all()):  # tiger.py:325-327
            raise ValueError("Messages' ts should be equal to last update ts when using left memory as msg source.")
        msgs = self.msg_transform(raw)
        old = self._mem(self.upd_src)[0][o]
        return outdated, self.updater(old, msgs), mts

    # ---- STEP 3 (temporal_agg_modules.py:29-83, 210-235) ---------------------
    def embed(self, reprs, local_index, nids3, ts3, l1_nids, l1_eids, l1_ts, hop2=None):
        """temporal_agg_modules.py:29-83.  hop2 = (nids, eids, ts) [Q*K, K] for n_layers == 2: the neighbours are then
        embedded first (with the LAST attention layer, fns[n_layers - depth], at the ROOT's query time, :63) and their
        embeddings take the place of reprs + node features in the keys of the first layer."""
        if len(nids3) > EMBED_CHUNK and self._drop(1) is None:
            # every centre is embedded independently of the others: large batches go through in slices so that
            # the [Q, K, 3d] key tensors of a 65 536-event batch do not have to exist at once
            K = l1_nids.shape[1]
            parts = [self.embed(reprs, local_index, nids3[a:a + EMBED_CHUNK], ts3[a:a + EMBED_CHUNK],
                                l1_nids[a:a + EMBED_CHUNK], l1_eids[a:a + EMBED_CHUNK], l1_ts[a:a + EMBED_CHUNK],
                                None if hop2 is None else tuple(x[a * K:(a + EMBED_CHUNK) * K] for x in hop2))
                     for a in range(0, len(nids3), EMBED_CHUNK)]
            return torch.cat(parts, 0)
        rows = lambda ids: reprs[_t(local_index[ids])] + self.node_feat(_t(ids))
        c = rows(nids3)
        if hop2 is None:
            nb = rows(l1_nids)
        else:
            K = l1_nids.shape[1]
            h2_n, h2_e, h2_t = hop2
            inner = self._attend('temporal_embedding_fn.fns.1.', rows(l1_nids.ravel()), ts3.repeat_interleave(K),
                                 rows(h2_n), h2_n, h2_e, h2_t)
            nb = inner.reshape(len(nids3), K, self.d)
        return self._attend('temporal_embedding_fn.fns.0.', c, ts3, nb, l1_nids, l1_eids, l1_ts)

    def _attend(self, pre, c, ts, nb, l_nids, l_eids, l_ts):
        """one TemporalAttention layer (temporal_agg_modules.py:52-81,210-235): centre rows c [n, d] at times ts,
        key node rows nb [n, K, d], edge ids / times of the keys, padding mask from l_nids == 0"""
        ln = _t(l_nids)
        ef = self.edge_feat(_t(l_eids))
        delta = ts[:, None] - _t(l_ts)
        kt = self.te(delta)
        qt = self.te(torch.zeros_like(delta[:, 0]))
        mask = (ln == 0)
        invalid = mask.all(1, keepdim=True)
        mask = mask.clone()
        mask[invalid.squeeze(1), -1] = False
        query = torch.cat([c, qt], 1).unsqueeze(0)
        kv = torch.cat([nb, ef, kt], 2).transpose(0, 1)
        h = mha(query, kv, kv, self.p[pre + 'mha_fn.q_proj_weight'], self.p[pre + 'mha_fn.k_proj_weight'],
                self.p[pre + 'mha_fn.v_proj_weight'], self.p[pre + 'mha_fn.in_proj_bias'],
                self.p[pre + 'mha_fn.out_proj.weight'], self.p[pre + 'mha_fn.out_proj.bias'],
                self.n_head, mask, drop=self._drop(1)).squeeze(0)
        h = h.masked_fill(invalid, 0.0)
        return merge_layer(h, c, self.p, pre + 'merger.')

    # ---- STEP 5 (tiger.py:422-442, memory.py:77-106) -------------------------
    def store_events(self, src, dst, ts, eids):
        s, dd = _t(src), _t(dst)
        mv, mt = self._mem(self.msg_src)
        sv, spt = mv[s].clone(), mt[s]
        dv, dpt = mv[dd].clone(), mt[dd]
        if (spt > ts).any() or (dpt > ts).any():
            raise ValueError('Events occur before the udpated memory.')
        pos = np.concatenate([src, dst])
        if self.has_msg[pos].any():  # memory.py:85-87
            raise ValueError(f'Node #{int(pos[self.has_msg[pos]][0])} has unused messages.')
        sv = sv + self.node_feat(s)
        dv = dv + self.node_feat(dd)
        ev = self.edge_feat(_t(eids))
        sm = torch.cat([sv, dv, ev, self.te(ts - spt)], 1)
        dm = torch.cat([dv, sv, ev, self.te(ts - dpt)], 1)
        ts2 = ts.repeat(2)
        ids, idx = select_latest_nids(pos, ts2.numpy())
        self.msg_vals[_t(ids)] = torch.cat([sm, dm], 0)[_t(idx)].detach()  # tiger.py:422 @torch.no_grad
        self.msg_ts[_t(ids)] = ts2[_t(idx)]
        self.has_msg[ids] = True

    # ---- the batch (tiger.py:174-290) ------------------------------------------
    def contrast_learning(self, src, dst, neg, ts, eids, cg: Dict[str, np.ndarray]):
        src, dst, neg, eids = (np.asarray(x, dtype=np.int64) for x in (src, dst, neg, eids))
        ts = _t(np.asarray(ts)).float()  # data_loader.py:92
        B = len(src)
        pos = np.concatenate([src, dst])
        nids3 = np.concatenate([src, dst, neg])
        involved = cg['involved']
        outdated, h_new, mts = self.consume(involved)  # STEP 1
        reprs = self.right_vals[_t(involved)].clone()  # STEP 2 (always the right memory)
        if len(outdated):
            reprs[_t(cg['local_index'][outdated])] = h_new
        hop2 = (cg['hop2_nids'], cg['hop2_eids'], cg['hop2_ts']) if 'hop2_nids' in cg else None
        h = self.embed(reprs, cg['local_index'], nids3, ts.repeat(3), cg['l1_nids'], cg['l1_eids'], cg['l1_ts'], hop2)
        if len(outdated):  # STEP 4
            upos, _ = select_latest_nids(pos, ts.repeat(2).numpy())
            where = np.searchsorted(outdated, upos)
            where[where >= len(outdated)] = 0
            sel = outdated[where] == upos
            if sel.any():
                ids = upos[sel]
                self.has_msg[ids] = False  # messages are consumed (memory.py:136)
                self._mem_set(self.right_vals, self.right_ts, _t(ids), h_new[_t(where[sel])], mts[_t(where[sel])])
        self.store_events(src, dst, ts, eids)  # STEP 5
        h_prev_left = self.left_vals[_t(pos)].clone()  # side quest, tiger.py:248-251
        h_prev_right = self.right_vals[_t(pos)].clone()
        h_left = h[:2 * B]  # STEP 6
        ts2 = ts.repeat(2)
        ids, idx = select_latest_nids(pos, ts2.numpy())
        self._mem_set(self.left_vals, self.left_ts, _t(ids), h_left[_t(idx)], ts2[_t(idx)])
        # STEP 7 (tiger.py:257-288): adjacent to the metric; kept for the score fixtures
        x, y, ny = h.reshape(3, B, self.d)
        hits = [_t(cg[k]) for k in ('src_hits', 'dst_hits', 'neg_src_hits', 'neg_dst_hits')]
        if self.hit_type == 'vec':
            xp, yp, xn, yn = (torch.cat([a, b], 1) for a, b in zip((x, y, x, ny), hits))
        elif self.hit_type in ('bin', 'count'):
            emb = self.p['hit_embedding.weight']
            red = (lambda t: t.max(1).values.long()) if self.hit_type == 'bin' else (lambda t: t.sum(1).long())
            xp, yp, xn, yn = (a + emb[red(b)] for a, b in zip((x, y, x, ny), hits))
        else:
            xp, yp, xn, yn = x, y, x, ny
        dp, dn = self._drop(2), self._drop(2, B * self.d)
        ps = merge_layer(xp, yp, self.p, 'score_fn.', None if dp is None else dp(B * self.d).reshape(B, self.d)).squeeze(1)
        ns = merge_layer(xn, yn, self.p, 'score_fn.', None if dn is None else dn(B * self.d).reshape(B, self.d)).squeeze(1)
        logits = torch.cat([ps, ns])
        labels = torch.cat([torch.ones_like(ps), torch.zeros_like(ns)])
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, labels)
        return dict(loss=loss, h_left=h_left, pos_scores=ps, neg_scores=ns,
                    h_prev_left=h_prev_left, h_prev_right=h_prev_right)

    # ---- training tail (tiger.py:547-592, train_self_supervised.py:143-171) ----
    def train_losses(self, src, dst, neg, ts, eids, cg, contrast_only=False):
        """contrast_and_mutual_learning with the parameters as autograd leaves.
        Returns (contrast_loss, mutual_loss) as differentiable torch scalars."""
        for v in self.p.values():
            v.requires_grad_(True)
            v.grad = None
        with torch.enable_grad():
            r = self.contrast_learning(src, dst, neg, ts, eids, cg)
            c_loss = r['loss']
            if contrast_only:  # tiger.py:570-572
                return c_loss, torch.zeros(())
            index = _t(cg['rd_index'])
            pos = np.concatenate([np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)])
            u_nids = pos[cg['rd_index']]
            u_ts = _t(np.asarray(ts)).float().repeat(2)[index]
            sl, sr, _ = self.restarter_forward(u_nids, u_ts.numpy(), cg)
            targets = torch.cat([r['h_prev_left'][index], r['h_prev_right'][index]], 0).detach()
            preds = torch.cat([sl, sr], 0)
            valid = torch.where(~(targets == 0).all(1))[0]
            m_loss = torch.nn.functional.mse_loss(preds[valid], targets[valid]) if len(valid) else torch.zeros(())
        return c_loss, m_loss

    def train_step(self, src, dst, neg, ts, eids, cg, *, lr, mutual_coef=1.0, contrast_only=False,
                   betas=(0.9, 0.999), eps=1e-8):
        """One optimisation step: losses, backward, torch.optim.Adam update (defaults, no weight
        decay).  Returns (contrast_loss, mutual_loss, grads)."""
        c_loss, m_loss = self.train_losses(src, dst, neg, ts, eids, cg, contrast_only)
        (c_loss + mutual_coef * m_loss).backward()
        grads = {k: (torch.zeros_like(v) if v.grad is None else v.grad.clone()) for k, v in self.p.items()}
        if not hasattr(self, 'adam'):
            self.adam = {k: [torch.zeros_like(v), torch.zeros_like(v), 0] for k, v in self.p.items()}
        with torch.no_grad():
            for k, v in self.p.items():
                if v.grad is None:  # Adam skips parameters without a gradient (their step count too)
                    continue
                g = v.grad
                self.adam[k][2] += 1
                m1, m2, t = self.adam[k]
                m1.mul_(betas[0]).add_(g, alpha=1 - betas[0])
                m2.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
                bc1, bc2 = 1 - betas[0] ** t, 1 - betas[1] ** t
                denom = (m2.sqrt() / np.sqrt(bc2)).add_(eps)
                v.addcdiv_(m1, denom, value=-lr / bc1)
        for v in self.p.values():
            v.requires_grad_(False)
            v.grad = None
        if self.dropout is not None:  # the library advances its mask counter once per training step
            self.dropout = (self.dropout[0], self.dropout[1], self.dropout[2] + 1)
        return float(c_loss.detach()), float(m_loss.detach()), grads

    # ---- streaming step: STEP 1-6 only, the benchmarked path -----------------
    def stream_step(self, src, dst, neg, ts, eids, cg):
        return self.contrast_learning(src, dst, neg, ts, eids, cg)['h_left']

    def flush_msg(self):  # tiger.py:444-455
        outdated, h_new, mts = self.consume(None)
        if len(outdated):
            self._mem_set(self.right_vals, self.right_ts, _t(outdated), h_new, mts)
            self.has_msg[outdated] = False

    # ---- restarters (restarters.py:36-114, 254-277) --------------------------
    def restarter_forward(self, nids: np.ndarray, ts: np.ndarray, cg: Optional[Dict] = None):
        nids = np.asarray(nids, dtype=np.int64)
        n = _t(nids)
        if self.restarter == 'static':
            if cg is None:  # restarters.py:265-270, float32 query timestamps
                prev_ts = _t(self.graph.get_history(nids, np.asarray(ts, dtype=F32), 1)[2][:, 0])
            else:
                prev_ts = _t(cg['rd_prev_ts'])
            return self.p['restarter_fn.left_emb.weight'][n], self.p['restarter_fn.right_emb.weight'][n], prev_ts
        if cg is None:  # restarters.py:67-76
            h_n, h_e, h_t, h_d = self.graph.get_history(nids, np.asarray(ts, dtype=F32), self.H)
            anon = anonymized_reindex(h_n)
        else:
            h_n, anon, h_e, h_t, h_d = (cg[k] for k in ('rd_hist_nids', 'rd_anon', 'rd_hist_eids',
                                                         'rd_hist_ts', 'rd_hist_dirs'))
        hn, he, ht, hd, an = _t(h_n), _t(h_e), _t(h_t), _t(h_d), _t(anon)
        d = self.d
        mask = (hn == 0)
        mask[:, -1] = False
        invalid = mask.all(1, keepdim=True)
        r = n.unsqueeze(1).repeat(1, hn.shape[1])
        s_n = r * hd + hn * (1 - hd)
        d_n = r * (1 - hd) + hn * hd
        tw, tp = self.p['restarter_fn.time_encoder.basis_freq'], self.p['restarter_fn.time_encoder.phase']
        full = torch.cat([self.node_feat(s_n), self.node_feat(d_n), self.p['restarter_fn.anony_emb.weight'][an],
                          self.edge_feat(he), time_encode(ht[:, -1:] - ht, tw, tp)], 2)
        dm = full.shape[2]
        # restarters.py:102-103: `last_event_feat` is a VIEW of full_vals that the next
        # line zeroes in place, so the merger really sees zeros (reference quirk, kept).
        full[:, -1, :dm - d] = 0.0
        last = full[:, -1, :dm - d]
        qkv = full.transpose(0, 1)
        w_in = self.p['restarter_fn.mha_fn.in_proj_weight']
        out = mha(qkv, qkv, qkv, w_in[:dm], w_in[dm:2 * dm], w_in[2 * dm:], self.p['restarter_fn.mha_fn.in_proj_bias'],
                  self.p['restarter_fn.mha_fn.out_proj.weight'], self.p['restarter_fn.mha_fn.out_proj.bias'],
                  self.n_head, mask, drop=self._drop(3))
        h_left = linear(torch.relu(out.mean(0)), self.p['restarter_fn.out_fn.weight'], self.p['restarter_fn.out_fn.bias'])
        dm_ = self._drop(4)
        h_right = merge_layer(h_left, last, self.p, 'restarter_fn.merger.',
                              None if dm_ is None else dm_(h_left.numel()).reshape(h_left.shape))
        return h_left.masked_fill(invalid, 0.0), h_right.masked_fill(invalid, 0.0), ht[:, -1]

    def restart(self, nids: np.ndarray, ts: np.ndarray):  # tiger.py:594-609 (mix == 0)
        nids = np.asarray(nids, dtype=np.int64)
        if len(nids) == 0:
            return
        self.has_msg[nids] = False
        hl, hr, pt = self.restarter_forward(nids, ts)
        if self.dropout is not None and self.restarter == 'seq' and torch.is_grad_enabled():
            # train() mode restart: the seq restarter's dropout was active; the library ticks its counter
            self.dropout = (self.dropout[0], self.dropout[1], self.dropout[2] + 1)
        self._mem_set(self.left_vals, self.left_ts, _t(nids), hl, pt, skip_check=True)
        self._mem_set(self.right_vals, self.right_ts, _t(nids), hr, pt, skip_check=True)

    def clear_msgs(self):  # memory.py:128-138 with nids=None
        self.has_msg[:] = False
