"""Mirror of tiger/model/restarters.py: SeqRestarter and StaticRestarter
(WalkRestarter cannot be constructed through init_model, init_utils.py:144-157)."""
import ctypes as C
from typing import Tuple

import torch
from torch import Tensor, nn

from .. import hip_ops
from .._lib import TgLinear, TgSeqRestarter, check, lib, ptr
from ..hip_ops import stream_ptr
from .basic_modules import MergeLayer
from .time_encoding import TimeEncode


class Restarter(nn.Module):
    def __init__(self, raw_feat_getter, graph):
        super().__init__()
        self.raw_feat_getter = raw_feat_getter
        self.graph = graph
        self.n_nodes = raw_feat_getter.n_nodes
        self.nfeat_dim = raw_feat_getter.nfeat_dim
        self.efeat_dim = raw_feat_getter.efeat_dim
        self.time_encoder = TimeEncode(dim=self.nfeat_dim)
        self.tfeat_dim = self.time_encoder.dim
        self.model_struct_fn = None  # set by TIGER: () -> TgModel of the owning model
        self.rng_fn = None           # set by TIGER: () -> device int64[2] dropout generator state

    def forward(self, nids: Tensor, ts: Tensor, computation_graph=None) -> Tuple[Tensor, Tensor, Tensor]:
        raise NotImplementedError


class SeqRestarter(Restarter):
    def __init__(self, raw_feat_getter, graph, *, hist_len: int = 20, n_head=2, dropout=0.1):
        super().__init__(raw_feat_getter, graph)
        self.hist_len = hist_len
        self.n_head = n_head
        self.anony_emb = nn.Embedding(self.hist_len + 1, self.nfeat_dim)
        self.d_model = self.nfeat_dim * 3 + self.efeat_dim + self.tfeat_dim
        self.mha_fn = nn.MultiheadAttention(self.d_model, n_head, dropout)
        self.out_fn = nn.Linear(self.d_model, self.nfeat_dim)
        self.merger = MergeLayer(self.nfeat_dim, self.d_model - self.tfeat_dim, self.nfeat_dim, self.nfeat_dim,
                                 dropout=dropout)

    def _struct(self) -> TgSeqRestarter:
        lin = lambda l: TgLinear(ptr(l.weight), ptr(l.bias))
        return TgSeqRestarter(self.hist_len, self.n_head, ptr(self.time_encoder.basis_freq),
                              ptr(self.time_encoder.phase), ptr(self.anony_emb.weight),
                              ptr(self.mha_fn.in_proj_weight), ptr(self.mha_fn.in_proj_bias),
                              lin(self.mha_fn.out_proj), lin(self.out_fn), lin(self.merger.fc1), lin(self.merger.fc2),
                              1 if self.raw_feat_getter.nfeats_all_zero() else 0, 0, ptr(self._ta_table()))

    def train(self, mode: bool = True):
        """Entering train() mode drops the tabulated anony_emb block: the library's optimizer (tg_adam_step, also inside
        replayed graphs) updates parameters through raw pointers, which torch's version counters - the table's key - do
        not see; whoever trains passes through train() first."""
        if mode:
            self._ta_cache = None
        return super().train(mode)

    def _ta_table(self):
        """Inference with fixed parameters on a zero node-feature table: the anony_emb block of the Q / K projection as a table
        T_a = anony_emb W[0:2dm, 2d:3d]^T (tg_seq_restarter.ta_cached), recomputed when either parameter changes (torch's
        version counters: any in-place update bumps them).  None in train() mode and with a non-zero node-feature table."""
        w, e = self.mha_fn.in_proj_weight, self.anony_emb.weight
        if self.training or torch.is_grad_enabled() or not w.is_cuda or not self.raw_feat_getter.nfeats_all_zero():
            return None
        key = (w.data_ptr(), w._version, e.data_ptr(), e._version)
        hit = getattr(self, '_ta_cache', None)
        if hit is None or hit[0] != key:
            d, dm = self.nfeat_dim, self.d_model
            from .dense import linear_forward
            lin = nn.Linear(d, 2 * dm, bias=True, device=w.device)
            with torch.no_grad():
                lin.weight.copy_(w[:2 * dm, 2 * d:3 * d])
                lin.bias.zero_()
                ta = linear_forward(lin, e.detach())
            hit = self._ta_cache = (key, ta.contiguous())
        return hit[1]

    def forward(self, nids: Tensor, ts: Tensor, computation_graph=None) -> Tuple[Tensor, Tensor, Tensor]:
        """restarters.py:51-114: surrogate h(t'-), h(t'+) and t' from the last hist_len events."""
        dev = nids.device
        if computation_graph is None:
            # history at restart time is queried with float32-rounded timestamps (restarters.py:69-70)
            h_n, h_e, h_t, h_d = self.graph.sample_device(nids, ts.float().double(), self.hist_len,
                                                         strategy='recent_edges')
            anon = hip_ops.anonymized_reindex(h_n)
        else:
            rd = computation_graph.restart_data
            h_n, anon, h_e, h_t, h_d = rd.hist_nids, rd.anonymized_ids, rd.hist_eids, rd.hist_ts, rd.hist_dirs
        n = nids.numel()
        d = self.nfeat_dim
        h_left = torch.empty(n, d, dtype=torch.float32, device=dev)
        h_right = torch.empty(n, d, dtype=torch.float32, device=dev)
        prev_ts = torch.empty(n, dtype=torch.float32, device=dev)
        if n == 0:
            return h_left, h_right, prev_ts
        m = self.model_struct_fn()
        r = self._struct()
        nbytes = int(lib.tg_restart_seq_workspace_bytes(C.byref(m), C.byref(r), n))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        nids = nids.long().contiguous()
        p = float(self.mha_fn.dropout)
        if self.training and p > 0 and self.rng_fn is not None:
            # train() mode: attention / merger dropout is active, also inside TIGER.restart (as in the reference)
            if float(self.merger.dropout.p) != p:
                raise NotImplementedError('one dropout probability for the restarter')
            check(lib.tg_restart_seq_fwd_train(C.byref(m), C.byref(r), n, ptr(nids), ptr(h_n), ptr(anon), ptr(h_e),
                                               ptr(h_t), ptr(h_d), ptr(h_left), ptr(h_right), ptr(prev_ts), p,
                                               ptr(self.rng_fn()), ptr(ws), nbytes, stream_ptr(dev)),
                  'tg_restart_seq_fwd_train')
            return h_left, h_right, prev_ts
        check(lib.tg_restart_seq_fwd(C.byref(m), C.byref(r), n, ptr(nids), ptr(h_n), ptr(anon), ptr(h_e), ptr(h_t),
                                     ptr(h_d), ptr(h_left), ptr(h_right), ptr(prev_ts), ptr(ws), nbytes,
                                     stream_ptr(dev)), 'tg_restart_seq_fwd')
        return h_left, h_right, prev_ts


class StaticRestarter(Restarter):
    def __init__(self, raw_feat_getter, graph):
        super().__init__(raw_feat_getter, graph)
        self.left_emb = nn.Embedding(self.n_nodes, self.nfeat_dim)
        self.right_emb = nn.Embedding(self.n_nodes, self.nfeat_dim)
        nn.init.zeros_(self.left_emb.weight)
        nn.init.zeros_(self.right_emb.weight)

    def forward(self, nids: Tensor, ts: Tensor, computation_graph=None) -> Tuple[Tensor, Tensor, Tensor]:
        """restarters.py:262-277"""
        if computation_graph is None:
            _, _, p_t, _ = self.graph.sample_device(nids, ts.float().double(), 1, strategy='recent_edges',
                                                    want_dirs=False)
            prev_ts = p_t[:, 0]
        else:
            prev_ts = computation_graph.restart_data.prev_ts  # [P, 1], as collated
        h_left = hip_ops.gather_rows(self.left_emb.weight.detach(), nids)
        h_right = hip_ops.gather_rows(self.right_emb.weight.detach(), nids)
        return h_left, h_right, prev_ts
