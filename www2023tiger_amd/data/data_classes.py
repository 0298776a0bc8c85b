"""Mirror of tiger/data/data_classes.py: per-batch containers.

Unlike the reference these are produced on the device by the collator, so `.to()`
and `.pin_memory()` only exist for loop compatibility (train_self_supervised.py:149,
DataLoader(pin_memory=True)).  The dense `local_index[n_nodes]` map of
data_classes.py:163-165 is replaced by (bitmap, rank); a dense view is built on
demand for callers that still ask for it.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor


class _Bundle:
    """Named tensors that move together."""
    _fields: Tuple[str, ...] = ()

    def __iter__(self):
        return (getattr(self, f) for f in self._fields)

    def to(self, device):
        for f in self._fields:
            setattr(self, f, getattr(self, f).to(device))
        return self

    def pin_memory(self):
        return self  # already device resident


class RestartData(_Bundle):
    pass


class SeqRestartData(RestartData):
    _fields = ('index', 'nids', 'ts', 'hist_nids', 'anonymized_ids', 'hist_eids', 'hist_ts', 'hist_dirs')

    def __init__(self, index, nids, ts, hist_nids, anonymized_ids, hist_eids, hist_ts, hist_dirs):
        self.index, self.nids, self.ts = index, nids, ts
        self.hist_nids, self.anonymized_ids, self.hist_eids = hist_nids, anonymized_ids, hist_eids
        self.hist_ts, self.hist_dirs = hist_ts, hist_dirs


class StaticRestartData(RestartData):
    _fields = ('index', 'nids', 'ts', 'prev_ts')

    def __init__(self, index, nids, ts, prev_ts):
        self.index, self.nids, self.ts, self.prev_ts = index, nids, ts, prev_ts


class HitData(_Bundle):
    _fields = ('src_hits', 'dst_hits', 'neg_src_hits', 'neg_dst_hits')

    def __init__(self, src_hits, dst_hits, neg_src_hits, neg_dst_hits):
        self.src_hits, self.dst_hits = src_hits, dst_hits
        self.neg_src_hits, self.neg_dst_hits = neg_src_hits, neg_dst_hits


class ComputationGraph:
    """layers[0] = (cat[src,dst,neg], None, None); layers[1] = (nids, eids, ts) [3B, K].
    The involved-node set is the bitmap; `involved`/`rank`/`n_involved` decode it.

    Built eagerly from collated pieces, or LAZILY from a collator and the batch arrays
    (`ComputationGraph.lazy`): the one-call training / evaluation steps collate on device
    themselves and only need `ts64` / `graph`, so the neighbourhood sample, the involved set,
    the restart data and the hit matrices are produced on first access only (the lazy-restart
    bookkeeping of the loops asks for `np_computation_graph_nodes`, nothing else)."""

    def __init__(self, layers: List[Tuple], bitmap: Tensor, rank: Tensor, involved: Tensor, n_involved: Tensor,
                 restart_data: Optional[RestartData], hit_data: Optional[HitData], n_nodes: int):
        self.n_nodes = n_nodes
        self._memory = (layers, bitmap, rank, involved, n_involved)
        self._restart = restart_data
        self._hit = hit_data
        self._pending = None
        self._count = None

    @classmethod
    def lazy(cls, collator, src_d: Tensor, dst_d: Tensor, neg_d: Tensor, ts64_d: Tensor):
        self = cls.__new__(cls)
        self.n_nodes = collator.n_nodes
        self._memory = self._restart = self._hit = None
        self._pending = (collator, src_d, dst_d, neg_d, ts64_d)
        self._count = None
        return self

    # ---- pieces, collated on first use ---------------------------------------------------
    def _mem(self):
        if self._memory is None:
            coll, s, d, n, t = self._pending
            layers, bitmap, comp = coll.collate_memory_nodes(torch.cat([s, d, n]), t.repeat(3))
            self._memory = (layers, bitmap, comp['rank'], comp['ids'], comp['count'])
        return self._memory

    layers = property(lambda self: self._mem()[0])
    bitmap = property(lambda self: self._mem()[1])
    rank = property(lambda self: self._mem()[2])
    involved = property(lambda self: self._mem()[3])
    n_involved = property(lambda self: self._mem()[4])

    @property
    def restart_data(self):
        if self._restart is None and self._pending is not None:
            coll, s, d, n, t = self._pending
            self._restart = coll.collate_restart_data(torch.cat([s, d]), t.repeat(2))
        return self._restart

    @property
    def hit_data(self):
        if self._hit is None and self._pending is not None:
            coll, s, d, n, t = self._pending
            self._hit = coll.collate_hit_data(s, d, n, t, self.layers[-1][0])  # the batch nodes' own neighbours
        return self._hit

    @property
    def device(self):
        return self._pending[1].device if self._pending is not None else self.bitmap.device

    @property
    def num_involved(self) -> int:
        if self._count is None:
            self._count = int(self.n_involved.item())  # host sync
        return self._count

    @property
    def computation_graph_nodes(self) -> Tensor:
        return self.involved[:self.num_involved]

    @property
    def np_computation_graph_nodes(self) -> np.ndarray:
        return self.computation_graph_nodes.cpu().numpy()

    @property
    def local_index(self) -> Tensor:
        idx = torch.zeros(self.n_nodes, dtype=torch.long, device=self.device)
        nodes = self.computation_graph_nodes
        idx[nodes] = torch.arange(len(nodes), device=self.device)
        return idx

    def to(self, device):
        return self

    def pin_memory(self):
        return self
