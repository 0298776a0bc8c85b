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
        if n_layers != 1:
            # every BASELINE config and the CLI default use one layer (init_utils.py:36)
            raise NotImplementedError('the HIP engine implements n_layers == 1')
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
        """temporal_agg_modules.py:29-83 for depth == n_layers == 1.  `involved_node_reprs`
        is indexed by the local index that (computation_graph.bitmap, rank) encode."""
        l1_n, l1_e, l1_t = computation_graph.layers[1]
        Q = center_nids.numel()
        d = involved_node_reprs.shape[1]
        out = torch.empty(Q, d, dtype=torch.float32, device=involved_node_reprs.device)
        nbytes = int(lib.tg_temporal_attn_workspace_bytes(C.byref(model_struct), Q))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=out.device)
        check(lib.tg_temporal_attn_fwd(C.byref(model_struct), Q, ptr(center_nids), ptr(ts), ptr(l1_n), ptr(l1_e),
                                       ptr(l1_t), ptr(involved_node_reprs), ptr(computation_graph.bitmap), ptr(rank),
                                       ptr(out), ptr(ws), nbytes, stream_ptr(out.device)), 'tg_temporal_attn_fwd')
        return out
