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
        if n_layers != 1:
            raise NotImplementedError('the HIP engine implements n_layers == 1')
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
        comp = hip_ops.unique_compact(None, self.n_nodes, cap, flags=flags)
        layers = [(nids3, None, None), (l_n, l_e, l_t)]
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

    def collate_arrays(self, src, dst, neg, ts, eids, labels=None):
        dev = self.graph.device
        ts64 = np.ascontiguousarray(ts, dtype=np.float64)
        s, d_, n_ = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.int64)) for x in (src, dst, neg))
        t_dev = torch.from_numpy(ts64).to(dev)
        s_d, d_d, n_d = s.to(dev), d_.to(dev), n_.to(dev)
        nids3 = torch.cat([s_d, d_d, n_d])
        layers, bitmap, comp = self.collate_memory_nodes(nids3, t_dev.repeat(3))
        restart = self.collate_restart_data(nids3[:2 * len(s)], t_dev.repeat(2))
        hit = self.collate_hit_data(s_d, d_d, n_d, t_dev, layers[1][0])
        cg = ComputationGraph(layers, bitmap, comp['rank'], comp['ids'], comp['count'], restart, hit, self.n_nodes)
        cg.ts64 = t_dev  # float64 event times for the fused training step (which collates on device itself)
        e = torch.from_numpy(np.ascontiguousarray(eids, dtype=np.int64))
        lab = torch.from_numpy(np.ascontiguousarray(labels, dtype=np.int64)) if labels is not None else None
        return s, d_, n_, torch.from_numpy(ts64).float(), e, lab, cg


class RandEdgeSampler:
    """data_loader.py:283-313 (host, numpy legacy RandomState stream)."""

    def __init__(self, src_list: np.ndarray, dst_list: np.ndarray, seed: Optional[int] = None):
        self.seed = seed
        self.rng = np.random.RandomState(self.seed)
        self.src_list = np.unique(src_list)
        self.dst_list = np.unique(dst_list)

    def sample(self, size: int):
        si = self.rng.randint(0, len(self.src_list), size)
        di = self.rng.randint(0, len(self.dst_list), size)
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

    def __len__(self):
        return len(self.ts)
