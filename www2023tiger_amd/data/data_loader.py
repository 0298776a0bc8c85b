"""Mirror of tiger/data/data_loader.py: `GraphCollator` (device collation),
`InteractionData`, `RandEdgeSampler`.

Per batch the reference makes 6 Python-loop sampler calls on the host
(data_loader.py:77-168).  Here one sampler launch covers cat[src,dst,neg] and
flags the involved nodes in the same kernel, the sorted unique set and the
local index come from a prefix-popcount over that bitmap, restart histories are a
second launch, and (under the default recent_edges strategy) the four hit
matrices are row blocks of the first launch's output.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import hip_ops
from .data_classes import ComputationGraph, HitData, SeqRestartData, StaticRestartData
from .graph import Graph


class GraphCollator:
    def __init__(self, graph: Graph, n_neighbors: int, n_layers: int, *, restarter: str = 'seq',
                 hist_len: Optional[int] = None, n_walks=None, walk_length=None, alpha: float = 0.0):
        if n_layers not in (1, 2):
            raise NotImplementedError('the HIP engine implements n_layers 1 and 2')
        if restarter not in ('seq', 'static'):
            raise NotImplementedError(restarter)  # 'walk' is unreachable from the CLI (init_utils.py:56-57)
        self.graph = graph
        self.n_nodes = graph.num_node
        self.n_neighbors = n_neighbors
        self.n_layers = n_layers
        self.restarter = restarter
        self.hist_len = hist_len

    # ---- pieces, device tensors in / out ------------------------------------------------
    def collate_memory_nodes(self, nids3: torch.Tensor, ts3: torch.Tensor):
        g, K = self.graph, self.n_neighbors
        dev = g.device
        flags = hip_ops.new_flags(self.n_nodes, dev)
        l_n, l_e, l_t, _ = g.sample_device(nids3, ts3, K, mark_flags=flags, want_dirs=False)
        cap = nids3.numel() * (K + 1)
        layers = [(nids3, None, None), (l_n, l_e, l_t)]
        if self.n_layers == 2:
            # data_loader.py:131: the next hop is sampled for every neighbour slot (padding slots included) at the
            # neighbour's own - float32 - timestamp; layers[2] are the batch nodes' neighbours, layers[1] the deepest hop
            h_n, h_e, h_t, _ = g.sample_device(l_n.reshape(-1).contiguous(), l_t.reshape(-1).double().contiguous(), K,
                                               mark_flags=flags, want_dirs=False)
            layers = [(nids3, None, None), (h_n, h_e, h_t), (l_n, l_e, l_t)]
            cap = nids3.numel() * (1 + K + K * K)
        comp = hip_ops.unique_compact(None, self.n_nodes, cap, flags=flags)
        return layers, comp['bitmap'], comp

    def collate_restart_data(self, pos: torch.Tensor, ts2: torch.Tensor):
        g = self.graph
        uniq, index = hip_ops.select_latest_nids(pos, ts2, self.n_nodes)  # float64 timestamps (data_loader.py:135)
        tu = ts2[index]
        if self.restarter == 'seq':
            h_n, h_e, h_t, h_d = g.sample_device(uniq, tu, self.hist_len, strategy='recent_edges')
            return SeqRestartData(index, uniq, tu.float(), h_n, hip_ops.anonymized_reindex(h_n), h_e, h_t, h_d)
        _, _, p_t, _ = g.sample_device(uniq, tu, 1, strategy='recent_edges', want_dirs=False)
        return StaticRestartData(index, uniq, tu.float(), p_t)  # prev_ts stays [P, 1] (data_loader.py:161-165)

    def collate_hit_data(self, src, dst, neg, ts, l1_nids: Optional[torch.Tensor]):
        B, K = src.numel(), self.n_neighbors
        if l1_nids is not None and self.graph.strategy == 'recent_edges':
            of_src, of_dst, of_neg = l1_nids[:B], l1_nids[B:2 * B], l1_nids[2 * B:]
        else:
            nb, _, _, _ = self.graph.sample_device(torch.cat([src, dst, neg]), ts.repeat(3), K,
                                                   strategy='recent_edges', want_dirs=False)
            of_src, of_dst, of_neg = nb[:B], nb[B:2 * B], nb[2 * B:]
        return HitData(hip_ops.hits(src, of_dst), hip_ops.hits(dst, of_src),
                       hip_ops.hits(src, of_neg), hip_ops.hits(neg, of_src))

    def __call__(self, batch: List[Tuple[int, int, int, float, int, int]]):
        src, dst, neg, ts, eids, labels = (np.array(x) for x in zip(*batch))
        return self.collate_arrays(src, dst, neg, ts, eids, labels)

    def collate_tensors(self, src, dst, neg, ts64, eids, labels=None):
        """collate_arrays for a batch that already lives on the collator's device (int64 ids, float64 times)"""
        cg = ComputationGraph.lazy(self, src, dst, neg, ts64)
        cg.ts64 = ts64
        cg.graph = self.graph if self.graph.strategy in ('recent_edges', 'recent_nodes') else None
        return src, dst, neg, ts64.float(), eids, labels, cg

    def collate_arrays(self, src, dst, neg, ts, eids, labels=None):
        """The batch as the reference's collate_fn returns it.  With the graph on a GPU the five id / time
        columns travel as ONE pinned, asynchronous transfer and are returned as device tensors (the loop's
        `.long().to(device)` / `.float().to(device)` are then no-ops): a pageable `.to(device)` per column
        blocks the host until the stream has drained, nine times per iteration in the reference's loop."""
        dev = self.graph.device
        ts64 = np.ascontiguousarray(ts, dtype=np.float64)
        lab = torch.from_numpy(np.ascontiguousarray(labels, dtype=np.int64)) if labels is not None else None
        if dev.type == 'cuda':
            B = len(ts64)
            stage = torch.empty(5, B, dtype=torch.int64, pin_memory=True)
            host = stage.numpy()
            host[0], host[1], host[2], host[3] = src, dst, neg, eids
            host[4] = ts64.view(np.int64)
            on_dev = stage.to(dev, non_blocking=True)
            s, d_, n_, e = on_dev[0], on_dev[1], on_dev[2], on_dev[3]
            t_dev = on_dev[4].view(torch.float64)
            s_d, d_d, n_d, t32 = s, d_, n_, t_dev.float()
        else:
            s, d_, n_, e = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.int64)) for x in (src, dst, neg, eids))
            t_dev = torch.from_numpy(ts64)
            s_d, d_d, n_d, t32 = s, d_, n_, t_dev.float()
        cg = ComputationGraph.lazy(self, s_d, d_d, n_d, t_dev)  # pieces are collated on first access
        cg.ts64 = t_dev  # float64 event times for the one-call steps (which collate on device themselves)
        # None: the one-call steps do not apply ('uniform' draws from the graph's random stream: the collator has drawn)
        cg.graph = self.graph if self.graph.strategy in ('recent_edges', 'recent_nodes') else None
        return s, d_, n_, t32, e, lab, cg


class RandEdgeSampler:
    """data_loader.py:283-313 (host, numpy legacy RandomState stream)."""

    def __init__(self, src_list: np.ndarray, dst_list: np.ndarray, seed: Optional[int] = None):
        self.seed = seed
        self._rng = np.random.RandomState(self.seed)
        self.src_list = np.unique(src_list)
        self.dst_list = np.unique(dst_list)
        self._dev = None        # device twin: (mt state uint32[625], src_list, dst_list) once sample_pairs_device ran
        self._dev_ahead = False  # the device copy of the generator state is ahead of the host RandomState

    @property
    def rng(self) -> np.random.RandomState:
        """the host generator; draws made on the device are folded back into it first (one stream, two homes)"""
        if self._dev_ahead:
            st = self._dev[0].cpu().numpy().view(np.uint32)
            name, _, _, has_gauss, cached = self._rng.get_state()
            self._rng.set_state((name, st[:624].copy(), int(st[624]), has_gauss, cached))
            self._dev_ahead = False
        return self._rng

    @rng.setter
    def rng(self, value):
        self._rng, self._dev_ahead = value, False
        if self._dev is not None:
            self._dev = (None,) + self._dev[1:]  # re-uploaded from the host state on the next device draw

    def sample_pairs_device(self, count: int, device):
        """`count` consecutive `sample(1)` calls ON THE DEVICE (tg_rand_edge_pairs): the same RandomState stream -
        host draws and device draws continue each other - returned as device tensors (src ids, dst ids), with no
        host work per batch beyond the launch."""
        from .._lib import check, lib, ptr
        device = torch.device(device)
        if self._dev is None or self._dev[1].device != device:
            self._dev = (None, torch.from_numpy(self.src_list.astype(np.int64)).to(device),
                         torch.from_numpy(self.dst_list.astype(np.int64)).to(device))
        if self._dev[0] is None or not self._dev_ahead:  # the host generator moved (or first use): upload its state
            _, key, pos, _, _ = self._rng.get_state()
            st = np.concatenate([key.astype(np.uint32), np.array([pos], dtype=np.uint32)])
            self._dev = (torch.from_numpy(st.view(np.int32).copy()).to(device),) + self._dev[1:]
        st, sl, dl = self._dev
        out_s = torch.empty(count, dtype=torch.int64, device=device)
        out_d = torch.empty(count, dtype=torch.int64, device=device)
        check(lib.tg_rand_edge_pairs(ptr(st), len(self.src_list), len(self.dst_list), count, ptr(sl), ptr(dl),
                                     ptr(out_s), ptr(out_d), hip_ops.stream_ptr(device)), 'tg_rand_edge_pairs')
        self._dev_ahead = True
        return out_s, out_d

    def sample(self, size: int):
        si = self.rng.randint(0, len(self.src_list), size)
        di = self.rng.randint(0, len(self.dst_list), size)
        return self.src_list[si], self.dst_list[di]

    def sample_pairs(self, count: int):
        """`count` consecutive `sample(1)` calls in one native call (same RandomState stream, same
        final state): the per-event draws of a training batch without a Python loop."""
        from .._lib import check, lib, ptr
        name, key, pos, has_gauss, cached = self.rng.get_state()
        st = np.concatenate([key.astype(np.uint32), np.array([pos], dtype=np.uint32)])
        si = np.empty(count, dtype=np.int64)
        di = np.empty(count, dtype=np.int64)
        check(lib.tg_rand_edge_pairs_host(ptr(st), len(self.src_list), len(self.dst_list), count, ptr(si), ptr(di)),
              'tg_rand_edge_pairs_host')
        self.rng.set_state((name, st[:624].copy(), int(st[624]), has_gauss, cached))
        return self.src_list[si], self.dst_list[di]

    def reset_random_state(self):
        self.rng = np.random.RandomState(self.seed)

    def pre_sample_neg_dsts(self, n_total: int, bs: int = 200) -> np.ndarray:
        self.reset_random_state()
        chunks, left = [], n_total
        while left > 0:
            take = min(bs, left)
            chunks.append(self.sample(take)[1])
            left -= take
        return np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.int64)


class InteractionData(torch.utils.data.Dataset):
    """data_loader.py:214-280"""

    def __init__(self, src, dst, ts, eids, labels, seed=0, eval=False, neg_dst=None):
        n = len(src)
        if not all(len(x) == n for x in (dst, ts, eids, labels)):
            raise AssertionError('all interaction arrays must have the same length')
        self.src, self.dst, self.ts, self.eids, self.labels = src, dst, ts, eids, labels
        self.eval, self.seed = eval, seed
        self._dev = None
        self.neg_dst = None
        self.neg_dst_sampler = RandEdgeSampler(src, dst, seed)
        if self.eval:
            self.neg_dst = neg_dst if neg_dst is not None else self.neg_dst_sampler.pre_sample_neg_dsts(n, bs=200)

    def get_subset(self, start, end):
        sl = slice(start, end)
        return InteractionData(self.src[sl], self.dst[sl], self.ts[sl], self.eids[sl], self.labels[sl], self.seed,
                               self.eval, self.neg_dst)

    def get_neg_dst_item(self, i) -> int:
        if self.eval:
            return self.neg_dst[i]
        return self.neg_dst_sampler.sample(1)[1].item()

    def __getitem__(self, i):
        return (self.src[i], self.dst[i], self.get_neg_dst_item(i), self.ts[i], self.eids[i], self.labels[i])

    def get_batch(self, lo: int, hi: int):
        """events [lo, hi) as arrays, negatives drawn exactly as `hi - lo` consecutive __getitem__ calls would"""
        sl = slice(lo, hi)
        neg = self.neg_dst[sl] if self.eval else self.neg_dst_sampler.sample_pairs(hi - lo)[1]
        return self.src[sl], self.dst[sl], neg, self.ts[sl], self.eids[sl], self.labels[sl]

    def to_device(self, device):
        """Keep the event columns resident on `device`: batches are then sliced there and the training negatives
        are drawn there (RandEdgeSampler.sample_pairs_device), so a training loop makes no per-batch host call
        on the input side (data_loader.py:246-251,291-294 without the per-event Python)."""
        device = torch.device(device)
        i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(device)
        self._dev = dict(device=device, src=i64(self.src), dst=i64(self.dst), eids=i64(self.eids), labels=i64(self.labels),
                         ts=torch.from_numpy(np.ascontiguousarray(self.ts, dtype=np.float64)).to(device),
                         neg=i64(self.neg_dst) if self.neg_dst is not None else None)
        return self

    def get_batch_device(self, lo: int, hi: int):
        """events [lo, hi) as device tensors (src, dst, neg, ts float64, eids, labels); negatives exactly those of
        `hi - lo` consecutive __getitem__ calls"""
        c = self._dev
        sl = slice(lo, hi)
        neg = c['neg'][sl] if self.eval else self.neg_dst_sampler.sample_pairs_device(hi - lo, c['device'])[1]
        return c['src'][sl], c['dst'][sl], neg, c['ts'][sl], c['eids'][sl], c['labels'][sl]

    def __len__(self):
        return len(self.ts)


class BatchLoader:
    """What the reference's loops need from `DataLoader(data, batch_size=bs, collate_fn=collator[, sampler=...])`
    - iteration over collated batches in order, `len()` - without one Python `__getitem__` per event:
    a batch is sliced as arrays, its negatives come from one native call that consumes the sampler's
    RandomState stream exactly as the per-event draws would.  `sampler` may be a ChunkSampler (a contiguous
    index range per epoch)."""

    def __init__(self, dataset: 'InteractionData', batch_size: int, collate_fn: GraphCollator, sampler=None):
        self.dataset, self.batch_size, self.collate_fn, self.sampler = dataset, batch_size, collate_fn, sampler

    def _range(self):
        if self.sampler is None:
            return 0, len(self.dataset)
        idx = list(iter(self.sampler))
        if idx and idx != list(range(idx[0], idx[0] + len(idx))):
            raise ValueError('BatchLoader needs a contiguous index range')
        return (idx[0], idx[0] + len(idx)) if idx else (0, 0)

    def __len__(self):
        lo, hi = self._range() if self.sampler is None else (0, len(self.sampler))
        return (hi - lo + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        lo, hi = self._range()
        on_device = getattr(self.dataset, '_dev', None) is not None
        for a in range(lo, hi, self.batch_size):
            b = min(a + self.batch_size, hi)
            if on_device:  # columns resident on the GPU, negatives drawn there: nothing crosses the bus
                yield self.collate_fn.collate_tensors(*self.dataset.get_batch_device(a, b))
            else:
                yield self.collate_fn.collate_arrays(*self.dataset.get_batch(a, b))


class ChunkSampler(torch.utils.data.Sampler):
    """data_loader.py:17-40: the time-chunk sampler of the reference's DDP recipe.  Rank r reads
    the contiguous index range [shift + r*len, shift + (r+1)*len), len = n // (world*bs) * bs;
    `shift` is drawn per epoch from the leftover events so that chunk boundaries move."""

    def __init__(self, n: int, rank: int, world_size: int, bs: int, seed: int = 0):
        self.n, self.rank, self.world_size, self.bs, self.seed = n, rank, world_size, bs, seed
        self.epoch = 0

    def __len__(self):
        return self.n // (self.world_size * self.bs) * self.bs

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __iter__(self):
        gen = torch.Generator()
        gen.manual_seed(self.seed + self.epoch)
        leftover = self.n % (self.world_size * self.bs)
        shift = int(torch.randint(0, leftover + 1, size=(), generator=gen))
        start = shift + len(self) * self.rank
        return iter(range(start, start + len(self)))


def _read_jodie_tables(root, name):
    import pathlib

    import pandas as pd
    base = pathlib.Path(root) / 'data'
    frame = pd.read_csv(base / f'ml_{name}.csv')
    feats = []
    for suffix in ('', '_node'):  # edge features, then node features; either file may be absent
        f = base / f'ml_{name}{suffix}.npy'
        feats.append(np.load(f) if f.exists() else None)
    return frame, feats[0], feats[1]


def load_jodie_data(name: str, train_seed: int, *, root='.', data_seed=2020, val_p=0.7, test_p=0.85):
    """data_loader.py:316-404: preprocessed JODIE files -> (nfeats, efeats, full, train, val, test,
    inductive_val, inductive_test).  Chronological split at the val_p / test_p time quantiles; 10 %
    of all nodes, drawn (python `random`, seed 2020 as TGAT/TGN) among the nodes seen after the
    validation time, are removed from training so that they are new at inference time; the
    inductive splits keep the events touching any node never seen in training."""
    import random
    frame, efeats, nfeats = _read_jodie_tables(root, name)
    src, dst, ts = frame.u.values, frame.i.values, frame.ts.values
    eids, labels = frame.idx.values, frame.label.values
    cols = (src, dst, ts, eids, labels)
    t_val, t_test = (float(q) for q in np.quantile(frame.ts, [val_p, test_p]))

    def subset(mask, seed, eval_mode):
        return InteractionData(*(c[mask] for c in cols), seed=seed, eval=eval_mode)

    all_nodes = set(src) | set(dst)
    late = ts > t_val
    # the draw depends on the iteration order of this set: build it exactly as a union of two sets
    seen_late = set(src[late]).union(set(dst[late]))
    hidden = set(random.Random(data_seed).sample(tuple(seen_late), int(0.1 * len(all_nodes))))
    touches_hidden = frame.u.isin(hidden).values | frame.i.isin(hidden).values
    train = subset((ts <= t_val) & ~touches_hidden, train_seed, False)
    trained_nodes = set(train.src) | set(train.dst)
    if trained_nodes & hidden:
        raise AssertionError('a held-out node leaked into the training split')
    unseen = list(all_nodes - trained_nodes)
    in_val, in_test = (ts > t_val) & (ts <= t_test), ts > t_test
    touches_unseen = np.isin(src, unseen) | np.isin(dst, unseen)
    return (nfeats, efeats, InteractionData(*cols),
            train, subset(in_val, 0, True), subset(in_test, 2, True),
            subset(in_val & touches_unseen, 1, True), subset(in_test & touches_unseen, 3, True))


def compute_delta_std(srcs: np.ndarray, dsts: np.ndarray, ts: np.ndarray) -> float:
    """data_loader.py:464-478: std of the time since each endpoint's previous event (first event: since 0)."""
    last = {}
    deltas = np.empty(2 * len(ts), dtype=np.float64)
    for k, (s, d, t) in enumerate(zip(srcs, dsts, ts)):
        deltas[2 * k], deltas[2 * k + 1] = t - last.get(s, 0), t - last.get(d, 0)
        last[s] = last[d] = t
    return float(np.std(deltas))


def is_sorted(x) -> bool:
    x = np.asarray(x)
    return bool(np.all(x[:-1] <= x[1:]))
