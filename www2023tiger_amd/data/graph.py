"""Temporal graph with a device-resident time-sorted CSR (T-CSR).

Mirror of tiger/data/graph.py (`Graph`): same constructor, `from_data`,
`num_node`, `sample_temporal_neighbor` and `get_history` signatures and return
types, but the per-query Python loop (graph.py:94-146, "Bottleneck! Total time
>50%") is replaced by HIP kernels over the T-CSR (csrc/tg_graph.hip).

The object is immutable after construction, so the collator thread and the main
thread may sample concurrently (each call allocates its own outputs and runs on
the calling thread's current stream).
"""
import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from .._lib import TgTcsr, check, lib, ptr
from ..hip_ops import stream_ptr


class Graph:
    def __init__(self, adj_list, strategy='recent_nodes', seed=None, alpha=0.0, device=None):
        """adj_list[n] = list of (neighbor, edge_index, timestamp, is_dst_flag), as built by
        the reference's data2adjlist (graph.py:226-241).  Prefer `from_data` /
        `from_arrays`, which never materialise Python tuples."""
        owner, nbr, eid, ts, flag = [], [], [], [], []
        for n, edges in enumerate(adj_list):
            for (o, e, t, f) in edges:
                owner.append(n)
                nbr.append(o)
                eid.append(e)
                ts.append(t)
                flag.append(f)
        owner = np.asarray(owner, dtype=np.int64)
        ts = np.asarray(ts, dtype=np.float64)
        order = np.lexsort((np.arange(len(owner)), ts, owner))  # stable sort by time inside a node (graph.py:32)
        eid = np.asarray(eid, dtype=np.int64)
        if len(eid) and (eid.min() < 0 or eid.max() > 0x7FFFFFFF):
            raise ValueError('edge ids must fit in 31 bits')
        self._init_common(len(adj_list), strategy, seed, alpha, device)
        packed = eid[order].astype(np.uint32) | (np.asarray(flag, dtype=np.uint32)[order] << np.uint32(31))
        self._host = (np.concatenate([[0], np.cumsum(np.bincount(owner, minlength=self.num_node))]).astype(np.int64),
                      ts[order], np.asarray(nbr, dtype=np.int64)[order].astype(np.int32), packed.view(np.int32))

    def _init_common(self, num_node, strategy, seed, alpha, device):
        self.num_node = int(num_node)
        self.strategy = strategy
        self.seed = seed
        self.alpha = alpha
        self.rng = np.random.RandomState(seed)  # graph.py:22; its state seeds the device MT19937
        self._device = torch.device(device) if device is not None else None
        self._dev = None  # device tensors, built / uploaded lazily
        self._mt = None
        self._events = None  # (src, dst, ts, eids) of from_arrays, the input of either builder
        self._host = None    # host T-CSR arrays (indptr, ts, nbr, eid), built on demand
        self._time_ordered = False

    @classmethod
    def from_arrays(cls, src, dst, ts, eids, strategy='recent_nodes', seed=None, max_node_id=None, device=None):
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        eids = np.ascontiguousarray(eids, dtype=np.int64)
        if max_node_id is None:
            max_node_id = int(max(src.max(), dst.max()))
        self = cls.__new__(cls)
        self._init_common(max_node_id + 1, strategy, seed, 0.0, device)
        E = len(src)
        if E and (min(src.min(), dst.min()) < 0 or max(src.max(), dst.max()) >= self.num_node):
            raise ValueError('node ids must lie in [0, num_node)')
        if E and (eids.min() < 0 or eids.max() > 0x7FFFFFFF):
            raise ValueError('edge ids must fit in 31 bits')
        self._events = (src, dst, ts, eids)
        # a time-ordered stream (every JODIE file) is built on the GPU; others take the host builder,
        # which also performs the reference's stable per-node sort by time (graph.py:32)
        self._time_ordered = bool(E < 2 or np.all(ts[1:] >= ts[:-1])) and 2 * E < 2 ** 32
        return self

    def _host_tcsr(self):
        if self._host is None:
            src, dst, ts, eids = self._events
            E = len(src)
            h = (np.empty(self.num_node + 1, dtype=np.int64), np.empty(2 * E, dtype=np.float64),
                 np.empty(2 * E, dtype=np.int32), np.empty(2 * E, dtype=np.int32))
            check(lib.tg_tcsr_build_host(E, ptr(src), ptr(dst), ptr(ts), ptr(eids), self.num_node, *(ptr(a) for a in h)),
                  'tg_tcsr_build_host')
            self._host = h
        return self._host

    _h_indptr = property(lambda self: self._host_tcsr()[0])
    _h_ts = property(lambda self: self._host_tcsr()[1])
    _h_nbr = property(lambda self: self._host_tcsr()[2])
    _h_eid = property(lambda self: self._host_tcsr()[3])

    def _build_on_device(self, dev):
        src, dst, ts, eids = (torch.from_numpy(a).to(dev) for a in self._events)
        E = src.numel()
        out = (torch.empty(self.num_node + 1, dtype=torch.int64, device=dev),
               torch.empty(2 * E, dtype=torch.float64, device=dev), torch.empty(2 * E, dtype=torch.int32, device=dev),
               torch.empty(2 * E, dtype=torch.int32, device=dev))
        nbytes = int(lib.tg_tcsr_build_device_workspace_bytes(E, self.num_node))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        check(lib.tg_tcsr_build_device(E, ptr(src), ptr(dst), ptr(ts), ptr(eids), self.num_node, *(ptr(t) for t in out),
                                       ptr(ws), nbytes, stream_ptr(dev)), 'tg_tcsr_build_device')
        torch.cuda.current_stream(dev).synchronize()  # the inputs and the workspace die with this scope
        return out

    @classmethod
    def from_data(cls, data, strategy='recent_nodes', seed=None, max_node_id=None, device=None):
        """`data` is an InteractionData-like object with .src/.dst/.ts/.eids arrays (graph.py:38-42)."""
        return cls.from_arrays(data.src, data.dst, data.ts, data.eids, strategy=strategy, seed=seed,
                               max_node_id=max_node_id, device=device)

    # ---- device residency ---------------------------------------------------------
    @property
    def device(self) -> torch.device:
        if self._device is None:
            self._device = torch.device('cuda', torch.cuda.current_device())
        return self._device

    def to(self, device):
        device = torch.device(device)
        if self._device != device:
            self._device, self._dev, self._mt = device, None, None
        return self

    def _tensors(self):
        if self._dev is None:
            dev = self.device
            if self._time_ordered and self._events is not None and self._host is None:
                self._dev = self._build_on_device(dev)  # tg_tcsr_build_device: stable radix sort on the owner id
            else:
                self._dev = tuple(torch.from_numpy(a).to(dev) for a in self._host_tcsr())
            self._struct = TgTcsr(self.num_node, self._dev[1].numel(), *[t.data_ptr() for t in self._dev])
        return self._dev

    @property
    def tcsr(self) -> TgTcsr:
        self._tensors()
        return self._struct

    def _mt_state(self) -> Tensor:
        if self._mt is None:
            _, key, pos, _, _ = self.rng.get_state()
            st = np.concatenate([key.astype(np.uint32), np.array([pos], dtype=np.uint32)]).view(np.int32)
            self._mt = torch.from_numpy(st.copy()).to(self.device)
        return self._mt

    # ---- sampling -------------------------------------------------------------------
    def sample_device(self, nids: Tensor, ts: Tensor, n_neighbors: int, strategy: Optional[str] = None,
                      mark_flags: Optional[Tensor] = None, want_dirs: bool = True
                      ) -> Tuple[Tensor, Tensor, Tensor, Optional[Tensor]]:
        """Device-tensor form of sample_temporal_neighbor: nids int64[Q], ts float64[Q]."""
        strategy = self.strategy if strategy is None else strategy
        g = self.tcsr
        dev = self.device
        nids = nids.to(dev, torch.int64).contiguous()
        ts = ts.to(dev, torch.float64).contiguous()  # float32 queries widen exactly (np.searchsorted does the same)
        Q, K = nids.numel(), int(n_neighbors)
        assert ts.numel() == Q
        o_n = torch.empty(Q, K, dtype=torch.int64, device=dev)
        o_e = torch.empty(Q, K, dtype=torch.int64, device=dev)
        o_t = torch.empty(Q, K, dtype=torch.float32, device=dev)
        o_d = torch.empty(Q, K, dtype=torch.int64, device=dev) if want_dirs else None
        s = stream_ptr(dev)
        if strategy == 'recent_edges':
            check(lib.tg_sample_recent_edges(C.byref(g), Q, ptr(nids), ptr(ts), K, ptr(o_n), ptr(o_e), ptr(o_t),
                                             ptr(o_d), ptr(mark_flags), s), 'tg_sample_recent_edges')
        elif strategy == 'recent_nodes':
            check(lib.tg_sample_recent_nodes(C.byref(g), Q, ptr(nids), ptr(ts), K, ptr(o_n), ptr(o_e), ptr(o_t),
                                             ptr(o_d), s), 'tg_sample_recent_nodes')
        elif strategy == 'uniform':
            check(lib.tg_sample_uniform(C.byref(g), Q, ptr(nids), ptr(ts), K, ptr(self._mt_state()), ptr(o_n),
                                        ptr(o_e), ptr(o_t), ptr(o_d), s), 'tg_sample_uniform')
        else:
            raise NotImplementedError(strategy)
        if mark_flags is not None and strategy != 'recent_edges':
            from ..hip_ops import flags_mark
            flags_mark(nids, mark_flags, self.num_node)
            flags_mark(o_n.reshape(-1), mark_flags, self.num_node)
        return o_n, o_e, o_t, o_d

    def sample_temporal_neighbor(self, nids: np.ndarray, ts: np.ndarray, n_neighbors: int = 20,
                                 strategy: Optional[str] = None
                                 ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """graph.py:67-148 - numpy in, numpy out: ([bs,K] int64 neighbours, [bs,K] int64 edge ids,
        [bs,K] float32 timestamps, [bs,K] int64 directions), left padded with zeros."""
        assert len(nids) == len(ts)
        n_t = torch.from_numpy(np.ascontiguousarray(nids, dtype=np.int64))
        t_t = torch.from_numpy(np.ascontiguousarray(ts, dtype=np.float64))
        out = self.sample_device(n_t, t_t, n_neighbors, strategy)
        return tuple(o.cpu().numpy() for o in out)

    def get_history(self, nids: np.ndarray, ts: np.ndarray, hist_len: int):
        """graph.py:150-155"""
        return self.sample_temporal_neighbor(nids, ts, n_neighbors=hist_len, strategy='recent_edges')
