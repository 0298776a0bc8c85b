"""Mirror of tiger/model/feature_getter.py (NumericalFeature)."""
from typing import Optional

import torch
from torch import Tensor, nn

from .. import hip_ops


class NumericalFeature(nn.Module):
    """Raw node / edge feature tables.  A missing table means zeros of width `dim`
    (feature_getter.py:81-85,95-99).  The HIP engine reads the tables in place; the
    get_* methods are the standalone lookups (tg_gather_rows)."""

    def __init__(self, nfeats: Optional[Tensor], efeats: Optional[Tensor], dim: int, *, use_tsfm: bool = False,
                 register_buffer: bool = True, device: torch.device = None):
        super().__init__()
        if use_tsfm:
            raise NotImplementedError('use_tsfm is never enabled by init_model (init_utils.py:137-139)')
        self.pin_mem = register_buffer
        self.device = device
        self.use_tsfm = use_tsfm
        self.out_dim = dim
        self.n_nodes = self.n_edges = None
        nd = ed = None
        if nfeats is not None:
            self.n_nodes, nd = nfeats.shape
        if efeats is not None:
            self.n_edges, ed = efeats.shape
        prep = lambda t: None if t is None else t.float().contiguous()
        if register_buffer:
            # buffers are non-persistent like the reference (not part of the state_dict); they move with .to(device)
            self.register_buffer('nfeats', prep(nfeats), persistent=False)
            self.register_buffer('efeats', prep(efeats), persistent=False)
        else:
            # --no_feat_buffer (feature_getter.py:41-47,86-87: tables stay on the CPU, rows are copied per lookup): here the
            # tables stay in PINNED host memory, which the GPU addresses directly - the kernels read the rows they need over
            # the host link, nothing is staged and `.to(device)` does not move them (plain attributes, as in the reference)
            pin = lambda t: None if t is None else (t.pin_memory() if torch.cuda.is_available() else t)
            self.nfeats, self.efeats = pin(prep(nfeats)), pin(prep(efeats))
        self.nfeat_dim = nd if nd else dim
        self.efeat_dim = ed if ed else dim

    def nfeats_all_zero(self) -> bool:
        """True when there is no node-feature table or every entry of it is zero (every JODIE data set).  Checked once per
        table version (one reduction); consumers may then skip the table's column blocks (tg_seq_restarter.nfeats_zero)."""
        t = self.nfeats
        if t is None:
            return True
        key = (t.data_ptr(), t._version, tuple(t.shape))
        hit = getattr(self, '_nz_cache', None)
        if hit is None or hit[0] != key:
            hit = (key, not bool(torch.count_nonzero(t).item()))
            self._nz_cache = hit
        return hit[1]

    def _lookup(self, table, ids, width):
        if table is None:
            return torch.zeros(*ids.shape, width, device=ids.device)
        return hip_ops.gather_rows(table, ids)  # a pinned host table is read in place by the kernel

    def get_node_embeddings(self, nids: Tensor) -> Tensor:
        return self._lookup(self.nfeats, nids, self.out_dim)

    def get_edge_embeddings(self, eids: Tensor) -> Tensor:
        return self._lookup(self.efeats, eids, self.out_dim)
