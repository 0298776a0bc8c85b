"""Mirror of tiger/model/temporal_agg_modules.py (computation-graph path only; the
deprecated on-the-fly sampling path, temporal_agg_modules.py:85-165, is not built)."""
import ctypes as C

import torch
from torch import Tensor, nn

from .._lib import check, lib, ptr
from ..hip_ops import stream_ptr
from .basic_modules import MergeLayer


class TemporalAttention(nn.Module):
    """Parameter container with the reference's names (mha_fn.q_proj_weight, ...,
    merger.fc1/fc2); executed by tg_temporal_attn_fwd."""

    def __init__(self, nfeat_dim, efeat_dim, tfeat_dim, n_head=2, dropout=0.1):
        super().__init__()
        self.n_head = n_head
        self.dropout = dropout
        self.query_dim = nfeat_dim + tfeat_dim
        self.key_dim = nfeat_dim + efeat_dim + tfeat_dim
        self.merger = MergeLayer(self.query_dim, nfeat_dim, nfeat_dim, nfeat_dim)
        self.mha_fn = nn.MultiheadAttention(embed_dim=self.query_dim, num_heads=self.n_head, dropout=self.dropout,
                                            kdim=self.key_dim, vdim=self.key_dim)


class GraphAttnEmbedding(nn.Module):
    def __init__(self, raw_feat_getter, time_encoder, graph, n_neighbors=20, n_layers=2, n_head=2, dropout=0.1):
        super().__init__()
        if n_layers not in (1, 2):
            # every BASELINE config and the CLI default use one layer (init_utils.py:36); two run on the operator path
            raise NotImplementedError('the HIP engine implements n_layers 1 and 2')
        self.raw_feat_getter = raw_feat_getter
        self.time_encoder = time_encoder
        self.graph = graph
        self.n_neighbors = n_neighbors
        self.n_layers = n_layers
        self.n_head = n_head
        self.dropout = dropout
        self.fns = nn.ModuleList([TemporalAttention(
            nfeat_dim=raw_feat_getter.nfeat_dim, efeat_dim=raw_feat_getter.efeat_dim, tfeat_dim=time_encoder.dim,
            n_head=n_head, dropout=dropout) for _ in range(n_layers)])

    @property
    def device(self):
        return self.time_encoder.basis_freq.device

    def compute_embedding_with_computation_graph(self, involved_node_reprs: Tensor, center_nids: Tensor, ts: Tensor,
                                                 computation_graph, model_struct, rank: Tensor) -> Tensor:
        """temporal_agg_modules.py:29-83.  `involved_node_reprs` is indexed by the local index that
        (computation_graph.bitmap, rank) encode.  `model_struct`: the tg_model of the owning TIGE, or a callable
        layer -> tg_model (n_layers == 2: each attention layer has its own weights, fns[n_layers - depth]).
        Two layers: the K neighbours of every centre are embedded first - as Q*K centres over the hop-2 neighbours,
        with fns[1], at the ROOT's query time (:63) - and their embeddings are the node part of the keys of fns[0]."""
        cg = computation_graph
        struct = model_struct if callable(model_struct) else (lambda layer: model_struct)
        top_n, top_e, top_t = cg.layers[self.n_layers]
        Q, K = center_nids.numel(), self.n_neighbors
        d = involved_node_reprs.shape[1]
        dev = involved_node_reprs.device
        s = stream_ptr(dev)

        def attend(m, q, nids, qts, l_n, l_e, l_t, key_rows=None):
            out = torch.empty(q, d, dtype=torch.float32, device=dev)
            nbytes = int(lib.tg_temporal_attn_workspace_bytes(C.byref(m), q))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            if key_rows is None:
                check(lib.tg_temporal_attn_fwd(C.byref(m), q, ptr(nids), ptr(qts), ptr(l_n), ptr(l_e), ptr(l_t),
                                               ptr(involved_node_reprs), ptr(cg.bitmap), ptr(rank), ptr(out), ptr(ws),
                                               nbytes, s), 'tg_temporal_attn_fwd')
            else:
                check(lib.tg_temporal_attn_fwd_keys(C.byref(m), q, ptr(nids), ptr(qts), ptr(l_n), ptr(l_e), ptr(l_t),
                                                    ptr(involved_node_reprs), ptr(cg.bitmap), ptr(rank), ptr(key_rows),
                                                    ptr(out), ptr(ws), nbytes, s), 'tg_temporal_attn_fwd_keys')
            return out

        if self.n_layers == 1:
            return attend(struct(0), Q, center_nids, ts, top_n, top_e, top_t)
        hop_n, hop_e, hop_t = cg.layers[1]
        inner = attend(struct(1), Q * K, top_n.reshape(-1).contiguous(), ts.repeat_interleave(K).contiguous(),
                       hop_n, hop_e, hop_t)
        return attend(struct(0), Q, center_nids, ts, top_n, top_e, top_t, key_rows=inner)
